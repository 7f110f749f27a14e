// K1 -- fused quadtree spatial stage for gfx950.
//
// One workgroup owns one (frame t, root cell) pair and does, without leaving the CU:
//   pyramid of the root cell (quadtree_spatial_merger.py:9-86 of the reference), all parent<->child cosine
//   statistics (quadtree_builder.py:41-66), the top-down split decision (:68-74) and the emission of the
//   surviving nodes (:26-37, :73-74).
//
// Mapping to the hardware
//   * thread <-> VEC consecutive channels; a token row is read with one 16-byte (fp32 x4 / bf16 x8)
//     coalesced load per lane, every leaf token of the root cell is read from HBM exactly once;
//   * the whole root-cell pyramid (16 leaves + 4 mids + 1 top for a 3-level tree) lives in VGPRs in the
//     input dtype, so emitted features are stored straight from registers (no LDS staging, no re-read);
//   * the 41 partial dot products / norms per 3-level block are reduced across the wave with a
//     butterfly transpose-reduce (~44 shuffles), then across waves through LDS in a fixed order, so the
//     decisions are deterministic run to run;
//   * decisions are taken by the first few threads from the reduced statistics in LDS.
//
// Node features are written to the scratch matrix S at row  t*H*W + y1*W + x1  (the node's origin leaf), so
// "sorted by (t, y1, x1)" (quadtree_builder.py:198-203) is simply ascending row order: no sort is needed.
#include "sttm_kernels.h"

namespace sttm {

template <int D> struct TreeConst {
    static constexpr int NNODE = depth_base(D);          // nodes of a complete 4-ary tree with D levels
    static constexpr int NPAR = depth_base(D - 1);       // non-leaf nodes
    static constexpr int NLEAF = pow4(D - 1);
    static constexpr int NSTAT = NNODE + 4 * NPAR + D;   // norm2[node] | dot[parent][slot] | alias_norm2[level]
    __host__ __device__ static constexpr int dot_id(int parent, int k) { return NNODE + parent * 4 + k; }
    __host__ __device__ static constexpr int alias_id(int level) { return NNODE + 4 * NPAR + level; }
};

struct CellGeo {           // children of one cell: first child row/col and how many (1 or 2) per axis
    int rs, rc, cs, cc;
};
__device__ __forceinline__ CellGeo cell_children(const LevelDims& g, int lvl, int i, int j) {
    CellGeo c;
    c.rs = child_start(i, g.h[lvl + 1]);
    c.rc = child_count(i, g.h[lvl + 1]);
    c.cs = child_start(j, g.w[lvl + 1]);
    c.cc = child_count(j, g.w[lvl + 1]);
    return c;
}

// pool 4 child packs in slot order ((c0+c1)+c2)+c3 over the valid ones; avg divides by the child count.
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> pool4(const Pack<T, VEC> (&c)[4], const bool (&v)[4], bool sum_mode) {
    Pack<T, VEC> out;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) cnt += v[k] ? 1 : 0;
    const float scale = sum_mode ? 1.f : (cnt == 4 ? 0.25f : (cnt == 2 ? 0.5f : 1.f));
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        float s = c[0].get(e);                       // slot 0 is always valid for a valid parent
        if (v[1]) s += c[1].get(e);
        if (v[2]) s += c[2].get(e);
        if (v[3]) s += c[3].get(e);
        out.set(e, s * scale);
    }
    return out;
}

// A 3-level block (top, 4 mids, 16 leaves) held in registers; BL == 2 uses top + mid[] as its leaves,
// BL == 1 only `top`.
template <typename T, int VEC> struct BlockRegs {
    Pack<T, VEC> leaf[4][4];
    Pack<T, VEC> mid[4];
    Pack<T, VEC> top;
};
struct BlockGeo {
    bool v1[4];
    bool v2[4][4];
    int ci[4], cj[4];          // level+1 coordinates of the 4 children
    int li[4][4], lj[4][4];    // level+2 coordinates of the 16 grandchildren
};

template <int BL>
__device__ __forceinline__ void block_geometry(const LevelDims& g, int lvl, int i, int j, bool valid, BlockGeo& b) {
    if constexpr (BL == 1) return;
    const CellGeo c = cell_children(g, lvl, i, j);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int dy = k >> 1, dx = k & 1;
        b.v1[k] = valid && dy < c.rc && dx < c.cc;
        b.ci[k] = c.rs + dy;
        b.cj[k] = c.cs + dx;
        if constexpr (BL == 3) {
            CellGeo m = cell_children(g, lvl + 1, b.v1[k] ? b.ci[k] : 0, b.v1[k] ? b.cj[k] : 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ey = q >> 1, ex = q & 1;
                b.v2[k][q] = b.v1[k] && ey < m.rc && ex < m.cc;
                b.li[k][q] = m.rs + ey;
                b.lj[k][q] = m.cs + ex;
            }
        }
    }
}

struct SpatialCtx {
    const void* x;
    int64_t sT, sH, sW;
    int64_t c0;           // first channel of this thread
    bool active;          // c0 < C
    bool sum_mode;
};

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_leaf(const SpatialCtx& cx, int t, int y, int x, bool valid) {
    Pack<T, VEC> p;
    if (valid && cx.active) p = load_pack<T, VEC>(cx.x, (int64_t)t * cx.sT + (int64_t)y * cx.sH + (int64_t)x * cx.sW + cx.c0);
    else p.zero();
    return p;
}

// Load the leaves of a block whose top cell sits at level `lvl` and build mids + top.
template <typename T, int VEC, int BL>
__device__ __forceinline__ void load_and_pool_block(const SpatialCtx& cx, int t, int i, int j, bool valid,
                                                    const BlockGeo& b, BlockRegs<T, VEC>& r) {
    if constexpr (BL == 1) {
        r.top = load_leaf<T, VEC>(cx, t, i, j, valid);
    } else if constexpr (BL == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) r.mid[k] = load_leaf<T, VEC>(cx, t, b.ci[k], b.cj[k], b.v1[k]);
        r.top = pool4<T, VEC>(r.mid, b.v1, cx.sum_mode);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) r.leaf[k][q] = load_leaf<T, VEC>(cx, t, b.li[k][q], b.lj[k][q], b.v2[k][q]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (b.v1[k]) r.mid[k] = pool4<T, VEC>(r.leaf[k], b.v2[k], cx.sum_mode);
            else r.mid[k].zero();
        }
        r.top = pool4<T, VEC>(r.mid, b.v1, cx.sum_mode);
    }
    if (!valid) r.top.zero();
}

// Feature of cell (0,0) of level `lvl` (the alias target of invalid child slots, quirk Q1): pooled from
// scratch with the same arithmetic as the main tree.  LV = levels below `lvl` (0 = leaf level).
template <typename T, int VEC, int LV> struct AliasPool {
    static __device__ __forceinline__ Pack<T, VEC> run(const SpatialCtx& cx, const LevelDims& g, int t, int lvl, int i, int j) {
        const CellGeo c = cell_children(g, lvl, i, j);
        Pack<T, VEC> ch[4];
        bool v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = (k >> 1) < c.rc && (k & 1) < c.cc;
            if (v[k]) ch[k] = AliasPool<T, VEC, LV - 1>::run(cx, g, t, lvl + 1, c.rs + (k >> 1), c.cs + (k & 1));
            else ch[k].zero();
        }
        return pool4<T, VEC>(ch, v, cx.sum_mode);
    }
};
template <typename T, int VEC> struct AliasPool<T, VEC, 0> {
    static __device__ __forceinline__ Pack<T, VEC> run(const SpatialCtx& cx, const LevelDims&, int t, int, int i, int j) {
        return load_leaf<T, VEC>(cx, t, i, j, true);
    }
};

// Per-thread partial statistics of one block, transposed-reduced over the wave, accumulated into this
// wave's row of the LDS partial table.  `tn` = tree-node id of the block's top, `lvl` its level.
template <typename T, int VEC, int BL, int D>
__device__ __forceinline__ void block_stats(const BlockRegs<T, VEC>& r, const BlockGeo& b, bool valid, int tn, int lvl,
                                            const Pack<T, VEC> (&alias)[D], float* part, int lane) {
    using TC = TreeConst<D>;
    if constexpr (BL == 1) {
        float s[1] = {dot_pack(r.top, r.top)};
        const float tot = wave_reduce_many<1>(s, lane);
        if (lane == 0 && valid) part[tn] = tot;
    } else if constexpr (BL == 2) {
        float s[9];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[k] = dot_pack(r.mid[k], r.mid[k]);
            s[5 + k] = b.v1[k] ? dot_pack(r.top, r.mid[k]) : dot_pack(r.top, alias[lvl + 1]);
        }
        s[4] = dot_pack(r.top, r.top);
        const float tot = wave_reduce_many<9>(s, lane);
        bool ok;
        const int idx = wave_reduce_slot<9>(lane, ok);
        if (ok && valid) {
            int id;
            if (idx < 4) id = 4 * tn + 1 + idx;
            else if (idx == 4) id = tn;
            else id = TC::dot_id(tn, idx - 5);
            part[id] = tot;
        }
    } else {
        // two mids at a time (18 partials live instead of 41: keeps the kernel at 4 workgroups per CU)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float s[18];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int k = 2 * h + kk;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s[kk * 9 + q] = dot_pack(r.leaf[k][q], r.leaf[k][q]);
                    s[kk * 9 + 5 + q] = b.v2[k][q] ? dot_pack(r.mid[k], r.leaf[k][q]) : dot_pack(r.mid[k], alias[lvl + 2]);
                }
                s[kk * 9 + 4] = dot_pack(r.mid[k], r.mid[k]);
            }
            const float tot = wave_reduce_many<18>(s, lane);
            bool ok;
            const int idx = wave_reduce_slot<18>(lane, ok);
            if (ok && valid) {
                const int k = 2 * h + idx / 9, e = idx % 9;
                const int mid = 4 * tn + 1 + k;
                int id;
                if (e < 4) id = 4 * mid + 1 + e;
                else if (e == 4) id = mid;
                else id = TC::dot_id(mid, e - 5);
                part[id] = tot;
            }
        }
        float s[5];
        s[0] = dot_pack(r.top, r.top);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[1 + k] = b.v1[k] ? dot_pack(r.top, r.mid[k]) : dot_pack(r.top, alias[lvl + 1]);
        const float tot = wave_reduce_many<5>(s, lane);
        bool ok;
        const int idx = wave_reduce_slot<5>(lane, ok);
        if (ok && valid) part[idx == 0 ? tn : TC::dot_id(tn, idx - 1)] = tot;
    }
}

// store the emitted nodes of a block from registers; orow[node] = destination row in S or -1
template <typename T, int VEC, int BL>
__device__ __forceinline__ void block_store(const BlockRegs<T, VEC>& r, int tn, const int* orow, void* S, int C,
                                            const SpatialCtx& cx, bool skip_leaves) {
    if (!cx.active) return;
    if (BL > 1 || !skip_leaves) {
        const int row = orow[tn];
        if (row >= 0) store_pack<T, VEC>(S, (int64_t)row * C + cx.c0, r.top);
    }
    if constexpr (BL >= 2) {
        if (BL > 2 || !skip_leaves) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = orow[4 * tn + 1 + k];
                if (row >= 0) store_pack<T, VEC>(S, (int64_t)row * C + cx.c0, r.mid[k]);
            }
        }
    }
    if constexpr (BL == 3) {
        if (skip_leaves) return;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = orow[4 * (4 * tn + 1 + k) + 1 + q];
                if (row >= 0) store_pack<T, VEC>(S, (int64_t)row * C + cx.c0, r.leaf[k][q]);
            }
    }
}

// Locate tree node n below root cell (I, J): depth, cell coordinates, validity, and the ids on its path.
struct NodeLoc {
    int depth, i, j;
    bool valid;
    int path[kMaxLevels];     // node ids from the root (path[0] = 0) down to the node itself
    int slot[kMaxLevels];     // slot[d] = child slot taken to reach depth d (slot[0] unused)
    int pi[kMaxLevels], pj[kMaxLevels];   // cell coordinates along the path
};
__device__ __forceinline__ NodeLoc locate(const LevelDims& g, int I, int J, int n) {
    NodeLoc L;
    int d = 0;
    while (depth_base(d + 1) <= n) ++d;
    L.depth = d;
    const int rel = n - depth_base(d);
    L.i = I; L.j = J; L.valid = true;
    L.path[0] = 0; L.pi[0] = I; L.pj[0] = J;
    int id = 0;
    for (int s = 1; s <= d; ++s) {
        const int k = (rel >> (2 * (d - s))) & 3;
        const CellGeo c = cell_children(g, s - 1, L.i, L.j);
        if ((k >> 1) >= c.rc || (k & 1) >= c.cc) L.valid = false;
        L.i = c.rs + (k >> 1);
        L.j = c.cs + (k & 1);
        id = 4 * id + 1 + k;
        L.path[s] = id;
        L.slot[s] = k;
        if (!L.valid) { L.i = 0; L.j = 0; }
        L.pi[s] = L.i; L.pj[s] = L.j;
    }
    return L;
}

// F.cosine_similarity(p, c) = sum((p / max(|p|, eps)) * (c / max(|c|, eps))), eps = 1e-8, then `>= threshold` on
// the fp32 value (quadtree_builder.py:61-68).  With the fp32 sums dot, |p|^2, |c|^2 the test
//     fp32(dot / (max(|p|,eps) * max(|c|,eps))) >= thr
// is evaluated without sqrt/div: rounding to fp32 is monotonic, so it equals  sim >= lo  where lo is the smallest
// real that rounds to >= thr (`thr_lo`, computed on the host in double), and  sim >= lo  <=>
// dot*|dot| >= lo*|lo| * max(|p|^2, eps^2) * max(|c|^2, eps^2)   (the product of two fp32 values is exact in double).
__device__ __forceinline__ bool cosine_at_least(float dot, float n2a, float n2b, double lo_sq_signed) {
    const double a = fmax((double)n2a, 1e-16), b = fmax((double)n2b, 1e-16);
    const double d = (double)dot;
    return d * fabs(d) >= lo_sq_signed * (a * b);
}

// Same as locate() for a node whose depth DEPTH is known at compile time: every loop unrolls, the per-level
// arrays are register-allocated (the generic version indexes them dynamically and lands in scratch memory).
template <int DEPTH>
__device__ __forceinline__ NodeLoc locate_static(const LevelDims& g, int I, int J, int rel) {
    NodeLoc L;
    L.depth = DEPTH;
    L.i = I; L.j = J; L.valid = true;
    L.path[0] = 0; L.pi[0] = I; L.pj[0] = J; L.slot[0] = 0;
    int id = 0;
#pragma unroll
    for (int s = 1; s <= DEPTH; ++s) {
        const int k = (rel >> (2 * (DEPTH - s))) & 3;
        const CellGeo c = cell_children(g, s - 1, L.i, L.j);
        if ((k >> 1) >= c.rc || (k & 1) >= c.cc) L.valid = false;
        L.i = c.rs + (k >> 1);
        L.j = c.cs + (k & 1);
        id = 4 * id + 1 + k;
        L.path[s] = id;
        L.slot[s] = k;
        if (!L.valid) { L.i = 0; L.j = 0; }
        L.pi[s] = L.i; L.pj[s] = L.j;
    }
    return L;
}

// one (parent, slot) cosine test for parents of a compile-time depth
template <int D, int DEPTH>
__device__ __forceinline__ void parent_slot_test(const SpatialArgs& a, int I, int J, int rel, int k, const float* stat, int* stop) {
    using TC = TreeConst<D>;
    const NodeLoc L = locate_static<DEPTH>(a.dims, I, J, rel);
    if (!L.valid) return;
    const int p = depth_base(DEPTH) + rel;
    const CellGeo c = cell_children(a.dims, DEPTH, L.i, L.j);
    const bool vk = (k >> 1) < c.rc && (k & 1) < c.cc;
    const float n2c = vk ? stat[4 * p + 1 + k] : stat[TC::alias_id(DEPTH + 1)];
    if (!cosine_at_least(stat[TC::dot_id(p, k)], stat[p], n2c, a.thr_lo_sq)) stop[p] = 0;
}

// Phases 2-4 of one (frame, root cell) item, entered after the per-wave partial statistics are in LDS and a
// barrier: fixed-order cross-wave sums, every (parent, slot) cosine test + every node's inverse norm in parallel,
// then one thread per leaf position emits.  Ends with a barrier; orow[] then maps tree node -> row in S (or -1).
template <int D>
__device__ __forceinline__ void decide_and_emit(const SpatialArgs& a, int t, int I, int J, int item, const float* part,
                                                float* stat, int* stop, int* orow, int* lcount, double* inrm_l, int nwave) {
    using TC = TreeConst<D>;
    constexpr int NSTAT = TC::NSTAT;
    const LevelDims& g = a.dims;
    const int tid = threadIdx.x;
    const int HW = a.H * a.W;
    int* rc_list = a.rc_list + (int64_t)item * a.rc_stride;
    // ---- phase 2: cross-wave sum in a fixed order ------------------------------------------------------------
    for (int s2 = tid; s2 < NSTAT; s2 += blockDim.x) {
        float acc = 0.f;
        for (int w = 0; w < nwave; ++w) acc += part[w * NSTAT + s2];
        stat[s2] = acc;
    }
    __syncthreads();
    // ---- phase 3: every (parent, slot) cosine test and every node's inverse norm, one thread each, spread over
    //      the waves so the four SIMDs work side by side -----------------------------------------------------------
    for (int base = 0; base < 4 * TC::NPAR + TC::NNODE; base += blockDim.x) {
        const int e = base + (tid & 63) * nwave + (tid >> 6);  // lane-major: consecutive items land on different waves
        if (e < 4 * TC::NPAR) {
            const int p = e >> 2, k = e & 3;
            if constexpr (D >= 2) { if (p < depth_base(1)) parent_slot_test<D, 0>(a, I, J, p, k, stat, stop); }
            if constexpr (D >= 3) { if (p >= depth_base(1) && p < depth_base(2)) parent_slot_test<D, 1>(a, I, J, p - depth_base(1), k, stat, stop); }
            if constexpr (D >= 4) { if (p >= depth_base(2) && p < depth_base(3)) parent_slot_test<D, 2>(a, I, J, p - depth_base(2), k, stat, stop); }
            if constexpr (D >= 5) { if (p >= depth_base(3) && p < depth_base(4)) parent_slot_test<D, 3>(a, I, J, p - depth_base(3), k, stat, stop); }
        } else if (e < 4 * TC::NPAR + TC::NNODE) {
            const int n = e - 4 * TC::NPAR;
            inrm_l[n] = 1.0 / (sqrt((double)stat[n]) + 1e-8);     // temporal stage: x / (|x| + 1e-8)
        }
    }
    __syncthreads();
    // ---- phase 4: emission, one thread per leaf position of the root cell -------------------------------------
    // The leaf-level node learns its first stopped ancestor (or itself): that node is emitted; the leaf that is its
    // top-left descendant (all slots below it are 0) is the node's origin and writes the node's metadata.
    for (int q = tid; q < TC::NLEAF; q += blockDim.x) {
        const NodeLoc L = locate_static<D - 1>(g, I, J, q);
        if (!L.valid) continue;
        int da = D - 1;
#pragma unroll
        for (int d = D - 2; d >= 0; --d) {
            if (stop[L.path[d]]) da = d;                       // ends at the SHALLOWEST stopped ancestor
        }
        bool origin = true;
        int emit_node = L.path[D - 1], hi_i = L.i, hi_j = L.j;
#pragma unroll
        for (int d = D - 1; d >= 0; --d) {
            if (d > da) origin = origin && (L.slot[d] == 0);
            if (d == da) { emit_node = L.path[d]; hi_i = L.pi[d]; hi_j = L.pj[d]; }
        }
        const int leaf_row = t * HW + L.i * a.W + L.j;
        if (!origin) { a.meta[leaf_row] = 0u; continue; }
        // box of the emitted ancestor: top-left is this leaf; bottom-right follows last children down
#pragma unroll
        for (int m = 0; m < D - 1; ++m) {
            if (m >= da) {
                hi_i = child_start(hi_i, g.h[m + 1]) + child_count(hi_i, g.h[m + 1]) - 1;
                hi_j = child_start(hi_j, g.w[m + 1]) + child_count(hi_j, g.w[m + 1]) - 1;
            }
        }
        const int y2 = hi_i + 1, x2 = hi_j + 1;
        a.meta[leaf_row] = ((uint32_t)y2 << 16) | (uint32_t)x2;
        a.inrm[leaf_row] = inrm_l[emit_node];
        orow[emit_node] = leaf_row;
        const int pos = atomicAdd(lcount, 1);
        rc_list[1 + pos] = (L.i << 24) | (L.j << 16) | (y2 << 8) | x2;
    }
    __syncthreads();
    if (tid == 0) {
        rc_list[0] = *lcount;
        if (I == 0 && J == 0) a.frame_cnt[t] = 0;            // consumed (atomically) by the label kernels
    }
    if (item == 0 && tid < STTM_CNT_SLOTS) a.counts[tid] = 0;
    if (item == 0 && tid < 2) a.bar[tid] = 0;
}

template <typename T, int VEC, int BL, int UL, int MAXNT>
__global__ void __launch_bounds__(MAXNT, 4) k_spatial(SpatialArgs a) {
    constexpr int D = BL + UL;
    using TC = TreeConst<D>;
    constexpr int NSTAT = TC::NSTAT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nwave = blockDim.x / kWave;
    double* inrm_l = reinterpret_cast<double*>(smem_raw);         // [NNODE] 1 / (|node| + 1e-8)
    float* part = reinterpret_cast<float*>(inrm_l + TC::NNODE);   // [nwave][NSTAT]
    float* stat = part + nwave * NSTAT;                           // [NSTAT]
    int* stop = reinterpret_cast<int*>(stat + NSTAT);             // [NPAR] (>= 1)
    int* orow = stop + (TC::NPAR > 0 ? TC::NPAR : 1);             // [NNODE]
    int* lcount = orow + TC::NNODE;                               // [1] (+1 pad)

    const LevelDims& g = a.dims;
    const int R = g.h[0] * g.w[0];
    const int t = blockIdx.x / R;
    const int rcell = blockIdx.x % R;
    const int I = rcell / g.w[0], J = rcell % g.w[0];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;

    SpatialCtx cx;
    cx.x = a.x; cx.sT = a.sT; cx.sH = a.sH; cx.sW = a.sW;
    cx.c0 = (int64_t)tid * VEC;
    cx.active = cx.c0 < a.C;
    cx.sum_mode = a.sum_mode != 0;

    // ---- phase 0: issue the loads of the first block, then clear LDS while they fly ------------------
    BlockGeo bg;
    BlockRegs<T, VEC> regs;
    Pack<T, VEC> alias[D];
    Pack<T, VEC> upper[4];   // UL == 1: the 4 block tops ; UL == 2: the 4 level-1 cells
    Pack<T, VEC> root;
    (void)upper;

    for (int s = tid; s < nwave * NSTAT; s += blockDim.x) part[s] = 0.f;
    for (int s = tid; s < TC::NNODE; s += blockDim.x) orow[s] = -1;
    for (int s = tid; s < TC::NPAR; s += blockDim.x) stop[s] = 1;
    if (tid == 0) *lcount = 0;

    // alias features: cell (0,0) of level m is needed when some parent of level m-1 inside this root cell
    // has an invalid slot: level m has an odd height and we own row 0, or an odd width and we own col 0.
    bool need_alias[D];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        need_alias[m] = m >= 1 && (((g.h[m] & 1) && I == 0) || ((g.w[m] & 1) && J == 0));
        alias[m].zero();
    }
    if constexpr (D >= 2) if (need_alias[D - 1]) alias[D - 1] = AliasPool<T, VEC, 0>::run(cx, g, t, D - 1, 0, 0);
    if constexpr (D >= 3) if (need_alias[D - 2]) alias[D - 2] = AliasPool<T, VEC, 1>::run(cx, g, t, D - 2, 0, 0);
    if constexpr (D >= 4) if (need_alias[D - 3]) alias[D - 3] = AliasPool<T, VEC, 2>::run(cx, g, t, D - 3, 0, 0);
    if constexpr (D >= 5) if (need_alias[D - 4]) alias[D - 4] = AliasPool<T, VEC, 3>::run(cx, g, t, D - 4, 0, 0);
    __syncthreads();   // LDS cleared

    float* mypart = part + wave * NSTAT;
    // alias norms
#pragma unroll
    for (int m = 1; m < D; ++m) {
        if (need_alias[m]) {
            const float tot = wave_sum(dot_pack(alias[m], alias[m]));
            if (lane == 0) mypart[TC::alias_id(m)] = tot;
        }
    }

    // ---- phase 1: pyramid + statistics ---------------------------------------------------------------
    if constexpr (UL == 0) {
        block_geometry<BL>(g, 0, I, J, true, bg);
        load_and_pool_block<T, VEC, BL>(cx, t, I, J, true, bg, regs);
        if (a.dbg_mode == 2) {                  // ablation: loads + pooling only (one store keeps them alive)
            if (cx.active) store_pack<T, VEC>(a.S, (int64_t)blockIdx.x * a.C + cx.c0, regs.top);
            return;
        }
        block_stats<T, VEC, BL, D>(regs, bg, true, 0, 0, alias, mypart, lane);
        root = regs.top;
    } else if constexpr (UL == 1) {
        const CellGeo c0 = cell_children(g, 0, I, J);
        bool v0[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v0[k] = (k >> 1) < c0.rc && (k & 1) < c0.cc;
            const int bi = v0[k] ? c0.rs + (k >> 1) : 0, bj = v0[k] ? c0.cs + (k & 1) : 0;
            if (v0[k]) {
                block_geometry<BL>(g, 1, bi, bj, true, bg);
                load_and_pool_block<T, VEC, BL>(cx, t, bi, bj, true, bg, regs);
                block_stats<T, VEC, BL, D>(regs, bg, true, 1 + k, 1, alias, mypart, lane);
                upper[k] = regs.top;
            } else {
                upper[k].zero();
            }
        }
        root = pool4<T, VEC>(upper, v0, cx.sum_mode);
        float s[5];
        s[0] = dot_pack(root, root);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[1 + k] = v0[k] ? dot_pack(root, upper[k]) : dot_pack(root, alias[1]);
        const float tot = wave_reduce_many<5>(s, lane);
        bool ok;
        const int idx = wave_reduce_slot<5>(lane, ok);
        if (ok) mypart[idx == 0 ? 0 : TC::dot_id(0, idx - 1)] = tot;
    } else {
        const CellGeo c0 = cell_children(g, 0, I, J);
        bool v0[4];
#pragma unroll
        for (int k0 = 0; k0 < 4; ++k0) {
            v0[k0] = (k0 >> 1) < c0.rc && (k0 & 1) < c0.cc;
            upper[k0].zero();
            if (v0[k0]) {
                const int ui = c0.rs + (k0 >> 1), uj = c0.cs + (k0 & 1);
                const CellGeo c1 = cell_children(g, 1, ui, uj);
                Pack<T, VEC> bt[4];
                bool v1[4];
#pragma unroll
                for (int k1 = 0; k1 < 4; ++k1) {
                    v1[k1] = (k1 >> 1) < c1.rc && (k1 & 1) < c1.cc;
                    bt[k1].zero();
                    if (v1[k1]) {
                        const int bi = c1.rs + (k1 >> 1), bj = c1.cs + (k1 & 1);
                        block_geometry<BL>(g, 2, bi, bj, true, bg);
                        load_and_pool_block<T, VEC, BL>(cx, t, bi, bj, true, bg, regs);
                        block_stats<T, VEC, BL, D>(regs, bg, true, 4 * (1 + k0) + 1 + k1, 2, alias, mypart, lane);
                        bt[k1] = regs.top;
                    }
                }
                upper[k0] = pool4<T, VEC>(bt, v1, cx.sum_mode);
                float s[5];
                s[0] = dot_pack(upper[k0], upper[k0]);
#pragma unroll
                for (int k1 = 0; k1 < 4; ++k1) s[1 + k1] = v1[k1] ? dot_pack(upper[k0], bt[k1]) : dot_pack(upper[k0], alias[2]);
                const float tot = wave_reduce_many<5>(s, lane);
                bool ok;
                const int idx = wave_reduce_slot<5>(lane, ok);
                if (ok) mypart[idx == 0 ? (1 + k0) : TC::dot_id(1 + k0, idx - 1)] = tot;
            }
        }
        root = pool4<T, VEC>(upper, v0, cx.sum_mode);
        float s[5];
        s[0] = dot_pack(root, root);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[1 + k] = v0[k] ? dot_pack(root, upper[k]) : dot_pack(root, alias[1]);
        const float tot = wave_reduce_many<5>(s, lane);
        bool ok;
        const int idx = wave_reduce_slot<5>(lane, ok);
        if (ok) mypart[idx == 0 ? 0 : TC::dot_id(0, idx - 1)] = tot;
    }
    __syncthreads();

    if (a.dbg_mode == 1) return;
    decide_and_emit<D>(a, t, I, J, blockIdx.x, part, stat, stop, orow, lcount, inrm_l, nwave);

    // ---- phase 5: store the emitted features ----------------------------------------------------------------
    if constexpr (UL == 0) {
        block_store<T, VEC, BL>(regs, 0, orow, a.S, a.C, cx, a.leaves_in_x != 0);
    } else if constexpr (UL == 1) {
        if (cx.active && orow[0] >= 0) store_pack<T, VEC>(a.S, (int64_t)orow[0] * a.C + cx.c0, root);
        if (!stop[0]) {
            const CellGeo c0 = cell_children(g, 0, I, J);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool v = (k >> 1) < c0.rc && (k & 1) < c0.cc;
                if (!v) continue;
                const int tn = 1 + k;
                if (stop[tn]) {      // the block top itself is the emitted node: still in `upper`
                    if (cx.active && orow[tn] >= 0) store_pack<T, VEC>(a.S, (int64_t)orow[tn] * a.C + cx.c0, upper[k]);
                } else {             // something below the top is emitted: re-read the block (L2-resident)
                    const int bi = c0.rs + (k >> 1), bj = c0.cs + (k & 1);
                    block_geometry<BL>(g, 1, bi, bj, true, bg);
                    load_and_pool_block<T, VEC, BL>(cx, t, bi, bj, true, bg, regs);
                    block_store<T, VEC, BL>(regs, tn, orow, a.S, a.C, cx, a.leaves_in_x != 0);
                }
            }
        }
    } else {
        if (cx.active && orow[0] >= 0) store_pack<T, VEC>(a.S, (int64_t)orow[0] * a.C + cx.c0, root);
        if (!stop[0]) {
            const CellGeo c0 = cell_children(g, 0, I, J);
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0) {
                const bool v0 = (k0 >> 1) < c0.rc && (k0 & 1) < c0.cc;
                if (!v0) continue;
                const int un = 1 + k0;
                if (stop[un]) {
                    if (cx.active && orow[un] >= 0) store_pack<T, VEC>(a.S, (int64_t)orow[un] * a.C + cx.c0, upper[k0]);
                    continue;
                }
                const int ui = c0.rs + (k0 >> 1), uj = c0.cs + (k0 & 1);
                const CellGeo c1 = cell_children(g, 1, ui, uj);
#pragma unroll
                for (int k1 = 0; k1 < 4; ++k1) {
                    const bool v1 = (k1 >> 1) < c1.rc && (k1 & 1) < c1.cc;
                    if (!v1) continue;
                    const int bi = c1.rs + (k1 >> 1), bj = c1.cs + (k1 & 1);
                    block_geometry<BL>(g, 2, bi, bj, true, bg);
                    load_and_pool_block<T, VEC, BL>(cx, t, bi, bj, true, bg, regs);
                    block_store<T, VEC, BL>(regs, 4 * un + 1 + k1, orow, a.S, a.C, cx, a.leaves_in_x != 0);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Pipelined variant for 3-level trees (the production configurations: 14x14 / 27x27 / 13x24 at root_level 1).
// Persistent workgroups walk the (frame, root cell) items with a stride of gridDim.x and keep TWO items in
// registers: while item i is pooled, reduced, decided and stored, the 16 leaf rows of item i+stride are already
// in flight.  The plain kernel alternates "everyone loads" / "everyone computes" chip-wide (2048 workgroups in two
// residency rounds); here every CU has loads outstanding all the time.  All HBM loads of an item (leaves + the
// alias leaves of row-0 / col-0 items) are issued together, so no later load forces a wait on the prefetch.
// ---------------------------------------------------------------------------------------------------
template <typename T, int VEC> struct Item3 {
    Pack<T, VEC> leaf[4][4];
    Pack<T, VEC> araw[4];      // leaves under level-1 cell (0,0): the alias sources
};
struct Item3Geo {
    int t, I, J;
    BlockGeo bg;
    bool need1, need2;         // this item has parents with invalid slots at level 0 / level 1
    bool av[4];                // validity of the 4 children of level-1 cell (0,0)
};

__device__ __forceinline__ void item3_geometry(const LevelDims& g, int item, Item3Geo& q) {
    const int R = g.h[0] * g.w[0];
    q.t = item / R;
    const int rc = item - q.t * R;
    q.I = rc / g.w[0]; q.J = rc - q.I * g.w[0];
    block_geometry<3>(g, 0, q.I, q.J, true, q.bg);
    q.need1 = ((g.h[1] & 1) && q.I == 0) || ((g.w[1] & 1) && q.J == 0);
    q.need2 = ((g.h[2] & 1) && q.I == 0) || ((g.w[2] & 1) && q.J == 0);
    const CellGeo c = cell_children(g, 1, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) q.av[k] = (k >> 1) < c.rc && (k & 1) < c.cc;
}

template <typename T, int VEC>
__device__ __forceinline__ void item3_load(const SpatialCtx& cx, const Item3Geo& q, Item3<T, VEC>& r) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) r.leaf[k][e] = load_leaf<T, VEC>(cx, q.t, q.bg.li[k][e], q.bg.lj[k][e], q.bg.v2[k][e]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r.araw[k] = load_leaf<T, VEC>(cx, q.t, k >> 1, k & 1, (q.need1 && q.av[k]) || (q.need2 && k == 0));
}

template <typename T, int VEC, int MAXNT>
__global__ void __launch_bounds__(MAXNT, 2) k_spatial3_pipe(SpatialArgs a) {
    constexpr int D = 3;
    using TC = TreeConst<D>;
    constexpr int NSTAT = TC::NSTAT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nwave = blockDim.x / kWave;
    double* inrm_l = reinterpret_cast<double*>(smem_raw);
    float* part = reinterpret_cast<float*>(inrm_l + TC::NNODE);
    float* stat = part + nwave * NSTAT;
    int* stop = reinterpret_cast<int*>(stat + NSTAT);
    int* orow = stop + TC::NPAR;
    int* lcount = orow + TC::NNODE;

    const LevelDims& g = a.dims;
    const int total = a.T * g.h[0] * g.w[0];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    SpatialCtx cx;
    cx.x = a.x; cx.sT = a.sT; cx.sH = a.sH; cx.sW = a.sW;
    cx.c0 = (int64_t)tid * VEC;
    cx.active = cx.c0 < a.C;
    cx.sum_mode = a.sum_mode != 0;
    float* mypart = part + wave * NSTAT;

    auto process = [&](const Item3Geo& q, Item3<T, VEC>& it, int item) {
        __syncthreads();                               // the previous item's readers of LDS are done
        for (int s = tid; s < nwave * NSTAT; s += blockDim.x) part[s] = 0.f;
        for (int s = tid; s < TC::NNODE; s += blockDim.x) orow[s] = -1;
        for (int s = tid; s < TC::NPAR; s += blockDim.x) stop[s] = 1;
        if (tid == 0) *lcount = 0;
        // pyramid
        BlockRegs<T, VEC> r;
        Pack<T, VEC> alias[D];
        alias[0].zero();
        alias[2] = it.araw[0];
        if (q.need1) alias[1] = pool4<T, VEC>(it.araw, q.av, cx.sum_mode); else alias[1].zero();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r.leaf[k][e] = it.leaf[k][e];
            if (q.bg.v1[k]) r.mid[k] = pool4<T, VEC>(r.leaf[k], q.bg.v2[k], cx.sum_mode); else r.mid[k].zero();
        }
        r.top = pool4<T, VEC>(r.mid, q.bg.v1, cx.sum_mode);
        __syncthreads();                               // LDS cleared
        if (q.need1) { const float tot = wave_sum(dot_pack(alias[1], alias[1])); if (lane == 0) mypart[TC::alias_id(1)] = tot; }
        if (q.need2) { const float tot = wave_sum(dot_pack(alias[2], alias[2])); if (lane == 0) mypart[TC::alias_id(2)] = tot; }
        block_stats<T, VEC, 3, D>(r, q.bg, true, 0, 0, alias, mypart, lane);
        __syncthreads();
        decide_and_emit<D>(a, q.t, q.I, q.J, item, part, stat, stop, orow, lcount, inrm_l, nwave);
        block_store<T, VEC, 3>(r, 0, orow, a.S, a.C, cx, a.leaves_in_x != 0);
    };

    Item3<T, VEC> A, B;
    Item3Geo qa, qb;
    int item = blockIdx.x;
    if (item >= total) return;
    item3_geometry(g, item, qa);
    item3_load<T, VEC>(cx, qa, A);
    while (true) {
        int next = item + gridDim.x;
        bool more = next < total;
        if (more) { item3_geometry(g, next, qb); item3_load<T, VEC>(cx, qb, B); }
        process(qa, A, item);
        if (!more) break;
        item = next;
        next = item + gridDim.x;
        more = next < total;
        if (more) { item3_geometry(g, next, qa); item3_load<T, VEC>(cx, qa, A); }
        process(qb, B, item);
        if (!more) break;
        item = next;
    }
}

// ---------------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------------
template <typename T, int VEC, int BL, int UL>
static hipError_t launch_bl_ul(const SpatialArgs& a, int grid, int nt, hipStream_t stream) {
    constexpr int D = BL + UL;
    using TC = TreeConst<D>;
    const int nwave = nt / kWave;
    const size_t smem = sizeof(double) * TC::NNODE + sizeof(float) * ((size_t)nwave * TC::NSTAT + TC::NSTAT) +
                        sizeof(int) * ((TC::NPAR > 0 ? TC::NPAR : 1) + TC::NNODE + 4);
    if (nt <= 256) hipLaunchKernelGGL((k_spatial<T, VEC, BL, UL, 256>), dim3(grid), dim3(nt), smem, stream, a);
    else if (nt <= 512) hipLaunchKernelGGL((k_spatial<T, VEC, BL, UL, 512>), dim3(grid), dim3(nt), smem, stream, a);
    else hipLaunchKernelGGL((k_spatial<T, VEC, BL, UL, 1024>), dim3(grid), dim3(nt), smem, stream, a);
    return hipGetLastError();
}

template <typename T, int VEC>
static hipError_t launch_pipe3(const SpatialArgs& a, int items, int nt, hipStream_t stream) {
    using TC = TreeConst<3>;
    const int nwave = nt / kWave;
    const size_t smem = sizeof(double) * TC::NNODE + sizeof(float) * ((size_t)nwave * TC::NSTAT + TC::NSTAT) +
                        sizeof(int) * (TC::NPAR + TC::NNODE + 4);
    // two resident workgroups per CU at <= 512 threads (256 VGPRs each); one above
    int grid = nt <= 512 ? 512 : 256;
    if (grid > items) grid = items;
    if (nt <= 256) hipLaunchKernelGGL((k_spatial3_pipe<T, VEC, 256>), dim3(grid), dim3(nt), smem, stream, a);
    else if (nt <= 512) hipLaunchKernelGGL((k_spatial3_pipe<T, VEC, 512>), dim3(grid), dim3(nt), smem, stream, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

template <typename T, int VEC>
static hipError_t launch_depth(const SpatialArgs& a, int grid, int nt, hipStream_t stream) {
    if (a.dims.n_level == 3 && nt <= 512 && grid >= 1024 && a.pipeline) return launch_pipe3<T, VEC>(a, grid, nt, stream);
    switch (a.dims.n_level) {
        case 1: return launch_bl_ul<T, VEC, 1, 0>(a, grid, nt, stream);
        case 2: return launch_bl_ul<T, VEC, 2, 0>(a, grid, nt, stream);
        case 3: return launch_bl_ul<T, VEC, 3, 0>(a, grid, nt, stream);
        case 4: return launch_bl_ul<T, VEC, 3, 1>(a, grid, nt, stream);
        case 5: return launch_bl_ul<T, VEC, 3, 2>(a, grid, nt, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_spatial(const SpatialArgs& a, int dtype, int vec, int nt, hipStream_t stream) {
    const int grid = a.T * a.dims.h[0] * a.dims.w[0];
    if (dtype == STTM_F32) {
        if (vec == 4) return launch_depth<float, 4>(a, grid, nt, stream);
        if (vec == 2) return launch_depth<float, 2>(a, grid, nt, stream);
        if (vec == 1) return launch_depth<float, 1>(a, grid, nt, stream);
    } else if (dtype == STTM_BF16) {
        if (vec == 8) return launch_depth<bf16_t, 8>(a, grid, nt, stream);
        if (vec == 4) return launch_depth<bf16_t, 4>(a, grid, nt, stream);
        if (vec == 2) return launch_depth<bf16_t, 2>(a, grid, nt, stream);
    } else if (dtype == STTM_F16) {
        if (vec == 8) return launch_depth<f16_t, 8>(a, grid, nt, stream);
        if (vec == 4) return launch_depth<f16_t, 4>(a, grid, nt, stream);
        if (vec == 2) return launch_depth<f16_t, 2>(a, grid, nt, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace sttm
