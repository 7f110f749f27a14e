import os, time, torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), device_id=dev)
x = torch.zeros(160, dtype=torch.int32, device=dev)
for name, fn in [("all_reduce int32[160]", lambda: dist.all_reduce(x)), ("barrier(device_ids)", lambda: dist.barrier(device_ids=[0])),
                 ("barrier()", lambda: dist.barrier()), ("as_tensor list->dev", lambda: torch.as_tensor(list(range(160)), device=dev, dtype=torch.long))]:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e6:.1f} us")
dist.destroy_process_group()
