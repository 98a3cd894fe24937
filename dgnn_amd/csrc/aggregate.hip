// Edge-filtered mean aggregation, forward and backward (general CSR, any width).
//
//   forward : a_i = ( sum_{k in seg(i)} x_src[src_k] * phi_k ) / max(deg_i,1)
//   phi_k   = We . A_e + be (FE > 0, recomputed in registers, never written unless asked)
//           | phi[e]        (FE == 0, "given": Updated variant / 2-layer edge MLP)
//           | 1             (FE < 0, lin_e is None)
//
// Mapping: one wavefront owns one destination at a time; the 64 lanes own CPL consecutive channels
// each (a [c_in] row is one coalesced 256/512-byte access).  Everything indexed by the destination
// (rowptr, src, eid, the 20 edge features) is wave-uniform, so it is read through the scalar cache
// into SGPRs and the filter MLP is 20 v_fma with an SGPR operand per channel -- no LDS, no shuffles.
// Wide layers are tiled over blockIdx.y in chunks of 64*CPL channels.
//
// Numerics: multiply and add are rounded separately (__fmul_rn/__fadd_rn) and the segment is summed
// in plan order starting from 0, i.e. the order of torch_scatter's CPU scatter_add_ that the
// reference runs (surfaceNetStaticEdgeFilters.py:80 -> aggr='mean').  Division is IEEE.
#include "common.h"
#include "reduce_common.h"

namespace {

// row fragments of CPL consecutive channels; T = float (fp32 storage) or uint16_t (bf16 storage: values are widened to fp32
// on load and rounded to nearest-even bf16 on store -- all arithmetic in between is fp32)
__device__ __forceinline__ uint16_t f2bf(float v) {
    typedef __bf16 bf1 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const bf1 h = __builtin_convertvector(f2{v, 0.f}, bf1);
    return (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xFFFFu);
}
__device__ __forceinline__ float bf2f(uint32_t bits16) { return __builtin_bit_cast(float, bits16 << 16); }

template <int CPL, typename T>
struct Vec;
template <>
struct Vec<1, float> {
    float v[1];
    __device__ __forceinline__ void load(const float* p) { v[0] = *p; }
    __device__ __forceinline__ void store(float* p) const { *p = v[0]; }
};
template <>
struct Vec<2, float> {
    float v[2];
    __device__ __forceinline__ void load(const float* p) {
        float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x;
        v[1] = t.y;
    }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
};
template <>
struct Vec<1, uint16_t> {
    float v[1];
    __device__ __forceinline__ void load(const uint16_t* p) { v[0] = bf2f(*p); }
    __device__ __forceinline__ void store(uint16_t* p) const { *p = f2bf(v[0]); }
};
template <>
struct Vec<2, uint16_t> {
    float v[2];
    __device__ __forceinline__ void load(const uint16_t* p) {
        const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
        v[0] = bf2f(t & 0xFFFFu);
        v[1] = __builtin_bit_cast(float, t & 0xFFFF0000u);
    }
    __device__ __forceinline__ void store(uint16_t* p) const {
        *reinterpret_cast<uint32_t*>(p) = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    }
};

template <int CPL, int FE, typename T>
__global__ void __launch_bounds__(256) k_agg_fwd(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                                 const int32_t* __restrict__ eid, int64_t n_dst,
                                                 const T* __restrict__ x, int64_t ldx, int c_in,
                                                 const float* __restrict__ ea, int64_t lde,
                                                 const float* __restrict__ We, const float* __restrict__ be,
                                                 const T* __restrict__ phi, int64_t ldphi,
                                                 T* __restrict__ phi_out, int64_t ldphi_out,
                                                 T* __restrict__ a, int64_t lda) {
    constexpr int NW = FE > 0 ? FE : 1;
    const int lane = lane_id();
    const int c0 = (blockIdx.y * 64 + lane) * CPL;
    const bool on = c0 < c_in;  // c_in is a multiple of CPL on this path
    float w[CPL][NW], b[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        b[j] = 0.f;
#pragma unroll
        for (int f = 0; f < NW; ++f) w[j][f] = 0.f;
        if (FE > 0 && on) {
            b[j] = be[c0 + j];
#pragma unroll
            for (int f = 0; f < NW; ++f) w[j][f] = We[(int64_t)(c0 + j) * FE + f];
        }
    }
    const int64_t wave = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave; d < n_dst; d += nwaves) {
        const int beg = rowptr[d], end = rowptr[d + 1];
        float acc[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
        // (one edge per iteration: batching the 4 in-edges' loads, which pays in k_agg_bwd, measured 25-35 % SLOWER here -- the
        // kernel already runs 8 blocks per CU and the wave-uniform attribute rows travel through the scalar cache)
        for (int k = beg; k < end; ++k) {
            const int s = src[k];
            const int64_t e = eid ? eid[k] : k;
            Vec<CPL, T> xr;
            if (on) xr.load(x + (int64_t)s * ldx + c0);
            float p[CPL];
            if (FE > 0) {
                const float* ar = ea + e * lde;
                float A[NW];
#pragma unroll
                for (int f = 0; f < NW; ++f) A[f] = ar[f];
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    float t = b[j];
#pragma unroll
                    for (int f = 0; f < NW; ++f) t = __fmaf_rn(w[j][f], A[f], t);
                    p[j] = t;
                }
                if (phi_out && on) {
                    Vec<CPL, T> po;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) po.v[j] = p[j];
                    po.store(phi_out + e * ldphi_out + c0);
                }
            } else if (FE == 0) {
                Vec<CPL, T> pr;
                if (on) pr.load(phi + e * ldphi + c0);
#pragma unroll
                for (int j = 0; j < CPL; ++j) p[j] = pr.v[j];
            } else {
#pragma unroll
                for (int j = 0; j < CPL; ++j) p[j] = 1.f;
            }
            if (on) {
#pragma unroll
                for (int j = 0; j < CPL; ++j)
                    acc[j] = __fadd_rn(acc[j], FE < 0 ? xr.v[j] : __fmul_rn(xr.v[j], p[j]));
            }
        }
        if (on) {
            const float cnt = (float)max(end - beg, 1);
            Vec<CPL, T> o;
#pragma unroll
            for (int j = 0; j < CPL; ++j) o.v[j] = __fdiv_rn(acc[j], cnt);
            o.store(a + d * lda + c0);
        }
    }
}

// Backward over the transposed plan: one wavefront per SOURCE tet, so dx_src[s] is a register
// accumulation (no atomics) in ascending edge position -- the order of autograd's index_add_ for
// x_j = x.index_select(0, edge_index[0]).  Filter-weight gradients are kept per lane (CPL x (FE+1)
// registers), reduced across the block's 4 waves through LDS in a fixed order and written as one
// slab per block; k_reduce_slabs sums the slabs in block order (deterministic).
template <int CPL, int FE, typename T>
__global__ void __launch_bounds__(256) k_agg_bwd(const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_dst,
                                                 const int32_t* __restrict__ t_eid, int64_t n_src,
                                                 const int32_t* __restrict__ rowptr_dst, const T* __restrict__ x,
                                                 int64_t ldx, int c_in, const float* __restrict__ ea, int64_t lde,
                                                 const float* __restrict__ We, const float* __restrict__ be,
                                                 const T* __restrict__ phi, int64_t ldphi,
                                                 const T* __restrict__ da, int64_t ldda, T* __restrict__ dx,
                                                 int64_t lddx, T* __restrict__ dphi_out, int64_t lddphi,
                                                 float* __restrict__ slabs) {
    constexpr int NW = FE > 0 ? FE : 1;
    __shared__ float red[FE > 0 ? 4 * 64 * CPL * (FE + 1) : 1];
    const int lane = lane_id();
    const int c0 = (blockIdx.y * 64 + lane) * CPL;
    const bool on = c0 < c_in;
    float w[CPL][NW], b[CPL], gw[CPL][NW], gb[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        b[j] = 0.f;
        gb[j] = 0.f;
#pragma unroll
        for (int f = 0; f < NW; ++f) {
            w[j][f] = 0.f;
            gw[j][f] = 0.f;
        }
        if (FE > 0 && on) {
            b[j] = be[c0 + j];
#pragma unroll
            for (int f = 0; f < NW; ++f) w[j][f] = We[(int64_t)(c0 + j) * FE + f];
        }
    }
    const int wv = wave_id_uniform();
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t s = wave; s < n_src; s += nwaves) {
        const int beg = t_rowptr[s], end = t_rowptr[s + 1];
        Vec<CPL, T> xs;
#pragma unroll
        for (int j = 0; j < CPL; ++j) xs.v[j] = 0.f;
        if (on && dx) {
            // rows of x_src without out-edges still get dx = 0
        }
        if (on && end > beg) xs.load(x + s * ldx + c0);
        float acc[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
        // 4 out-edges per batch: indices first, then every row they point at, then the arithmetic in edge order (see k_agg_fwd)
        for (int k0 = beg; k0 < end; k0 += 4) {
            const int nk = min(4, end - k0);
            int dq[4];
            int64_t eq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kk = k0 + (q < nk ? q : 0);
                dq[q] = t_dst[kk];
                eq[q] = t_eid[kk];
            }
            float cntq[4];
            Vec<CPL, T> gq[4], prq[4];
            float A[4][NW];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cntq[q] = (float)max(rowptr_dst[dq[q] + 1] - rowptr_dst[dq[q]], 1);
                if (on) gq[q].load(da + (int64_t)dq[q] * ldda + c0);
                if (FE > 0) {
                    const float* ar = ea + eq[q] * lde;
#pragma unroll
                    for (int f = 0; f < NW; ++f) A[q][f] = ar[f];
                } else if (FE == 0) {
                    if (on) prq[q].load(phi + eq[q] * ldphi + c0);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= nk) break;
                float p[CPL];
                if (FE > 0) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        float t = b[j];
#pragma unroll
                        for (int f = 0; f < NW; ++f) t = __fmaf_rn(w[j][f], A[q][f], t);
                        p[j] = t;
                    }
                } else if (FE == 0) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) p[j] = prq[q].v[j];
                } else {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) p[j] = 1.f;
                }
                if (on) {
                    Vec<CPL, T> dph;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const float dm = __fdiv_rn(gq[q].v[j], cntq[q]);
                        acc[j] = __fadd_rn(acc[j], __fmul_rn(dm, p[j]));
                        dph.v[j] = __fmul_rn(dm, xs.v[j]);
                        if (FE > 0) {
                            gb[j] += dph.v[j];
#pragma unroll
                            for (int f = 0; f < NW; ++f) gw[j][f] = __fmaf_rn(dph.v[j], A[q][f], gw[j][f]);
                        }
                    }
                    if (FE == 0 && dphi_out) dph.store(dphi_out + eq[q] * lddphi + c0);
                }
            }
        }
        if (on && dx) {
            Vec<CPL, T> o;
#pragma unroll
            for (int j = 0; j < CPL; ++j) o.v[j] = acc[j];
            o.store(dx + s * lddx + c0);
        }
    }
    if (FE > 0) {
        // red[wave][channel-in-chunk][FE+1]
        float* mine = red + ((wv * 64 + lane) * CPL) * (FE + 1);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
#pragma unroll
            for (int f = 0; f < NW; ++f) mine[j * (FE + 1) + f] = gw[j][f];
            mine[j * (FE + 1) + FE] = gb[j];
        }
        __syncthreads();
        constexpr int PER = 64 * CPL * (FE + 1);
        float* slab = slabs + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * PER;
        for (int i = threadIdx.x; i < PER; i += 256)
            slab[i] = ((red[i] + red[PER + i]) + red[2 * PER + i]) + red[3 * PER + i];
    }
}

// out[chunk][i] = sum_b slabs[b][chunk][i] (written, not accumulated: callers need no zero fill); 16 outputs x 64 slices per block: slice s adds slabs s, s+64, ... in order,
// the 64 slice sums are then added in slice order (deterministic, and not a 1024-long dependent chain per output).
__global__ void __launch_bounds__(16 * RS_SLICES) k_reduce_slabs(SlabReduceDesc d) {
    __shared__ float red[RS_SLICES][17];
    reduce_slabs_block(d, blockIdx.x, red);
}

// =====================================================================================================================
// Chunked backward (default).  k_agg_bwd walks one source row at a time behind a chain of dependent scalar loads (row pointer ->
// edge indices -> in-degree, rows they point at): on the 4-hop training blocks (10^3..10^5 rows, a handful per wavefront) that
// chain is what the launch takes.  Here a wavefront owns a CHUNK of up to 16 consecutive rows: the row pointers of the chunk are
// one coalesced load (lane l holds t_rowptr[row0 + l]), the edge indices of up to 64 consecutive plan positions another (lane l
// holds position w + l), every lane finds the row of its position by counting row ends, and the rows the edges point at are
// fetched SLOTS edges at a time with the index broadcast by v_readlane.  The edge-attribute row of a slot is one 80-byte load
// (lane f holds feature f) and reaches the fma chains through v_readlane as an SGPR operand.  The arithmetic per channel is
// k_agg_bwd's in the same order (dx / dphi bit-identical; the filter-weight gradients are the same sums with rows dealt to lanes
// differently).  Measured on the blocks of a training batch: 200 -> 168 us over the four layers.  The same form of the FORWARD
// kernel was slower (99 vs 86 us: its attribute rows already travel through the scalar cache, and the v_readlane broadcasts
// cost more issue slots than the shorter chain saves) and was not kept.
// =====================================================================================================================
__device__ __forceinline__ int rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rlf(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

constexpr int CH_ROWS = 16;   // rows per chunk at most (row pointers in lanes 0 .. 16)
static bool no_dx_form() {
    static const bool on = !(getenv("DGNN_AGG_BWD_NODX") && getenv("DGNN_AGG_BWD_NODX")[0] == '0');
    return on;
}

// ADD: dx[row] = (the row's sum) + add[row] for row < n_add -- the `dx[:n_dst] += dz . Wi` of a conv layer's backward without a
// launch of its own (same rounding as the GEMM epilogue that used to accumulate into dx: fl(sum + addend)).  The chunk's addend rows are
// parked in the wavefront's part of `red` when the chunk starts (16 independent loads) so that no row waits for its own.
// DX = false (round 6): the first conv layer's backward -- its input is data, only dWe / dbe are wanted -- skips the filter's recomputation (phi is
// needed for dx alone: 20 of the 40 fused multiply-adds per channel and edge) and the dx bookkeeping.
template <int CPL, int FE, typename T, int SLOTS, bool ADD = false, bool DX = true>
__global__ void __launch_bounds__(256) k_agg_bwd_c(const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_dst,
                                                   const int32_t* __restrict__ t_eid, int64_t n_src, const int32_t* __restrict__ rowptr_dst,
                                                   const T* __restrict__ x, int64_t ldx, int c_in, const float* __restrict__ ea, int64_t lde,
                                                   const float* __restrict__ We, const float* __restrict__ be, const T* __restrict__ phi,
                                                   int64_t ldphi, const T* __restrict__ da, int64_t ldda, T* __restrict__ dx, int64_t lddx,
                                                   T* __restrict__ dphi_out, int64_t lddphi, float* __restrict__ slabs, int rows_per_chunk,
                                                   const T* __restrict__ add, int64_t ldadd, int64_t n_add, const T* __restrict__ dphi_ext, int mask_dx) {
    constexpr int NW = FE > 0 ? FE : 1;
    // ADD: the addend rows of a chunk are parked in the wavefront's part of `red` (the slab region when the filter is fused, a region of its own
    // in the given-phi form, which has no slabs).  dphi_ext (given-phi form): dphi_out[e] = dphi_e + dphi_ext[e] -- the gradient the NEXT layer sent
    // to this layer's phi through its own edge input (Updated variant), added where dphi is stored instead of by a launch over [E, c_in].
    constexpr int PARK = FE > 0 ? FE + 1 : CH_ROWS;
    static_assert(!ADD || PARK >= CH_ROWS, "the addend rows of a chunk fit the wavefront's region");
    __shared__ float red[FE > 0 ? 4 * 64 * CPL * (FE + 1) : (ADD ? 4 * 64 * CPL * CH_ROWS : 1)];
    const int lane = lane_id();
    const int c0 = (blockIdx.y * 64 + lane) * CPL;
    const bool on = c0 < c_in;
    float w[CPL][NW], b[CPL], gw[CPL][NW], gb[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        b[j] = 0.f;
        gb[j] = 0.f;
#pragma unroll
        for (int f = 0; f < NW; ++f) {
            w[j][f] = 0.f;
            gw[j][f] = 0.f;
        }
        if (FE > 0 && on) {
            b[j] = be[c0 + j];
#pragma unroll
            for (int f = 0; f < NW; ++f) w[j][f] = We[(int64_t)(c0 + j) * FE + f];
        }
    }
    const int wv = wave_id_uniform();
    const int RW = rows_per_chunk;
    const int64_t nchunks = (n_src + RW - 1) / RW;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t chunk = (int64_t)blockIdx.x * 4 + wv; chunk < nchunks; chunk += stride) {
        const int64_t rb = chunk * RW;
        const int nr = (int)(n_src - rb < RW ? n_src - rb : RW);
        const int rp = t_rowptr[rb + (lane < nr ? lane : nr)];
        const int beg0 = rl(rp, 0), endN = rl(rp, nr);
        int cur = 0;
        float acc[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
        float* const park = red + (wv * 64 + lane) * CPL * PARK;   // ADD: [row of the chunk][CPL] of this lane
        if (ADD && on) {
            Vec<CPL, T> av[CH_ROWS];
#pragma unroll
            for (int r = 0; r < CH_ROWS; ++r) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) av[r].v[j] = 0.f;
                if (r < nr && rb + r < n_add) av[r].load(add + (rb + r) * ldadd + c0);
            }
#pragma unroll
            for (int r = 0; r < CH_ROWS; ++r)
#pragma unroll
                for (int j = 0; j < CPL; ++j) park[r * CPL + j] = av[r].v[j];
        }
        auto finish_until = [&](int r) {   // rows without out-edges get dx = 0 like the others get their sum
            if (!DX) return;
            while (cur < r) {
                if (on && dx) {
                    Vec<CPL, T> o;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) o.v[j] = (ADD && rb + cur < n_add) ? __fadd_rn(acc[j], park[cur * CPL + j]) : acc[j];
                    if (mask_dx) {   // the ReLU behind the layer below (x = relu(y_below) is this layer's input): dx * [x > 0], see k_agg_bwd_g
                        Vec<CPL, T> xm;
                        xm.load(x + (rb + cur) * ldx + c0);
#pragma unroll
                        for (int j = 0; j < CPL; ++j) o.v[j] = xm.v[j] > 0.f ? o.v[j] : 0.f;
                    }
                    o.store(dx + (rb + cur) * lddx + c0);
                }
#pragma unroll
                for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
                ++cur;
            }
        };
        for (int wpos = beg0; wpos < endN; wpos += 64) {
            const int nw = min(64, endN - wpos);
            int dv = 0, ev = 0, rowv = 0, cv = 1;
            if (lane < nw) {
                dv = t_dst[wpos + lane];
                ev = t_eid[wpos + lane];
                cv = max(rowptr_dst[dv + 1] - rowptr_dst[dv], 1);
            }
            for (int m = 1; m <= nr; ++m) rowv += (rl(rp, m) <= wpos + lane) ? 1 : 0;
            for (int k0 = 0; k0 < nw; k0 += SLOTS) {
                Vec<CPL, T> gq[SLOTS], xq[SLOTS], pr[SLOTS], pe[SLOTS];
                float Av[SLOTS];
#pragma unroll
                for (int j = 0; j < SLOTS; ++j) {
                    const int kk = k0 + j < nw ? k0 + j : nw - 1;
                    const int d = rl(dv, kk);
                    const int64_t e = rl(ev, kk);
                    const int64_t s = rb + rl(rowv, kk);
                    if (on) {
                        gq[j].load(da + (int64_t)d * ldda + c0);
                        xq[j].load(x + s * ldx + c0);
                    }
                    if (FE > 0) Av[j] = lane < FE ? ea[e * lde + lane] : 0.f;
                    if (FE == 0 && on) pr[j].load(phi + e * ldphi + c0);
                    if (FE == 0 && on && dphi_ext) pe[j].load(dphi_ext + e * lddphi + c0);
                }
#pragma unroll
                for (int j = 0; j < SLOTS; ++j) {
                    if (k0 + j >= nw) break;
                    finish_until(rl(rowv, k0 + j));
                    const float cnt = (float)rl(cv, k0 + j);
                    float p[CPL], af[NW];
                    if (FE > 0) {
#pragma unroll
                        for (int jj = 0; jj < CPL; ++jj) p[jj] = b[jj];
#pragma unroll
                        for (int f = 0; f < NW; ++f) {
                            af[f] = rlf(Av[j], f);
                            if (DX) {
#pragma unroll
                                for (int jj = 0; jj < CPL; ++jj) p[jj] = __fmaf_rn(w[jj][f], af[f], p[jj]);
                            }
                        }
                    } else if (FE == 0) {
#pragma unroll
                        for (int jj = 0; jj < CPL; ++jj) p[jj] = on ? pr[j].v[jj] : 0.f;
                    } else {
#pragma unroll
                        for (int jj = 0; jj < CPL; ++jj) p[jj] = 1.f;
                    }
                    if (on) {
                        Vec<CPL, T> dph;
#pragma unroll
                        for (int jj = 0; jj < CPL; ++jj) {
                            const float dm = __fdiv_rn(gq[j].v[jj], cnt);
                            if (DX) acc[jj] = __fadd_rn(acc[jj], __fmul_rn(dm, p[jj]));
                            dph.v[jj] = __fmul_rn(dm, xq[j].v[jj]);
                            if (FE > 0) {
                                gb[jj] += dph.v[jj];
#pragma unroll
                                for (int f = 0; f < NW; ++f) gw[jj][f] = __fmaf_rn(dph.v[jj], af[f], gw[jj][f]);
                            }
                        }
                        if (FE == 0 && dphi_out) {
                            if (dphi_ext) {
#pragma unroll
                                for (int jj = 0; jj < CPL; ++jj) dph.v[jj] = __fadd_rn(dph.v[jj], pe[j].v[jj]);
                            }
                            dph.store(dphi_out + (int64_t)rl(ev, k0 + j) * lddphi + c0);
                        }
                    }
                }
            }
        }
        finish_until(nr);
    }
    if (FE > 0) {
        float* mine = red + ((wv * 64 + lane) * CPL) * (FE + 1);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
#pragma unroll
            for (int f = 0; f < NW; ++f) mine[j * (FE + 1) + f] = gw[j][f];
            mine[j * (FE + 1) + FE] = gb[j];
        }
        __syncthreads();
        constexpr int PER = 64 * CPL * (FE + 1);
        float* slab = slabs + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * PER;
        for (int i = threadIdx.x; i < PER; i += 256)
            slab[i] = ((red[i] + red[PER + i]) + red[2 * PER + i]) + red[3 * PER + i];
    }
}

// =====================================================================================================================
// Lane-group form of the given-phi backward (the Updated variant's conv).  In the kernels above a lane is ONE channel (or two) and a
// wavefront instruction serves one edge: at 28-64 channels half the lanes idle and every edge costs three row loads of a few dozen
// bytes each -- the launch is bound by the number of memory instructions, not by bytes.  Here a lane owns 4 consecutive channels
// (one 16-byte / 8-byte load), G = 8 / 16 / 32 lanes form a row, and the 64 / G groups of a wavefront work on 64 / G SOURCE ROWS
// at once, each walking its own out-edges in ascending position with up to four edges' loads in flight.  Per channel the
// arithmetic and its order are k_agg_bwd's (dx / dphi bit-identical).  The chunk's row pointers and edge indices are coalesced
// loads as in k_agg_bwd_c; a group takes its edges' indices by ds_bpermute.  The addend (dx[row] += add[row]) is one more row load
// at the start of a row.
// =====================================================================================================================
template <typename T>
struct V4;
template <>
struct V4<float> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
    }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <>
struct V4<uint16_t> {
    float v[4];
    __device__ __forceinline__ void load(const uint16_t* p) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = bf2f(t.x & 0xFFFFu), v[1] = __builtin_bit_cast(float, t.x & 0xFFFF0000u);
        v[2] = bf2f(t.y & 0xFFFFu), v[3] = __builtin_bit_cast(float, t.y & 0xFFFF0000u);
    }
    __device__ __forceinline__ void store(uint16_t* p) const {
        *reinterpret_cast<uint2*>(p) = make_uint2((uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16));
    }
};

template <int G, typename T, bool ADD>
__global__ void __launch_bounds__(256) k_agg_bwd_g(const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_dst, const int32_t* __restrict__ t_eid,
                                                   int64_t n_src, const int32_t* __restrict__ rowptr_dst, const T* __restrict__ x, int64_t ldx, int c_in,
                                                   const T* __restrict__ phi, int64_t ldphi, const T* __restrict__ da, int64_t ldda, T* __restrict__ dx,
                                                   int64_t lddx, T* __restrict__ dphi_out, int64_t lddphi, int rows_per_chunk, const T* __restrict__ add,
                                                   int64_t ldadd, int64_t n_add, int mask_dx) {
    // mask_dx (round 6): dx[row] *= [x[row] > 0] at the store -- the ReLU between the layer below and this one (this layer's input x IS relu(y_below),
    // so the mask is in the row the kernel already holds): the Updated stack's k_relu_bwd launch per layer is gone.
    constexpr int R = 64 / G;
    const int lane = lane_id(), g = lane / G, c0 = 4 * (lane % G);
    const bool on = c0 < c_in;
    const int RW = rows_per_chunk;
    const int64_t nchunks = (n_src + RW - 1) / RW;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave_id_uniform(); chunk < nchunks; chunk += stride) {
        const int64_t rb = chunk * RW;
        const int nr = (int)(n_src - rb < RW ? n_src - rb : RW);
        const int rp = t_rowptr[rb + (lane < nr ? lane : nr)];
        const int beg0 = rl(rp, 0), ne = rl(rp, nr) - beg0;
        const bool inw = ne <= 64;      // the chunk's edge indices fit the wavefront: one coalesced load each, handed out by ds_bpermute
        int dv = 0, ev = 0, cv = 1;
        if (inw && lane < ne) {
            dv = t_dst[beg0 + lane];
            ev = t_eid[beg0 + lane];
            cv = max(rowptr_dst[dv + 1] - rowptr_dst[dv], 1);
        }
        for (int r0 = 0; r0 < nr; r0 += R) {
            const int r = r0 + g;
            const bool rv = r < nr;
            const int rc = rv ? r : nr - 1;
            const int b = __shfl(rp, rc), deg = rv ? __shfl(rp, rc + 1) - b : 0;
            const int64_t row = rb + rc;
            V4<T> xs, av;
#pragma unroll
            for (int j = 0; j < 4; ++j) xs.v[j] = av.v[j] = 0.f;
            if (on && (deg > 0 || (mask_dx && rv))) xs.load(x + row * ldx + c0);
            const bool has_add = ADD && rv && row < n_add;
            if (has_add && on) av.load(add + row * ldadd + c0);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = 0; __any(t0 < deg); t0 += 4) {
                int d[4], e[4];
                float cnt[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ok[u] = t0 + u < deg;
                    const int k = b + (ok[u] ? t0 + u : 0);
                    if (inw) {
                        int sl = k - beg0;
                        sl = sl < 0 ? 0 : (sl > 63 ? 63 : sl);
                        d[u] = __shfl(dv, sl);
                        e[u] = __shfl(ev, sl);
                        cnt[u] = (float)__shfl(cv, sl);
                    } else {
                        d[u] = e[u] = 0;
                        cnt[u] = 1.f;
                        if (ok[u]) {
                            d[u] = t_dst[k];
                            e[u] = t_eid[k];
                            cnt[u] = (float)max(rowptr_dst[d[u] + 1] - rowptr_dst[d[u]], 1);
                        }
                    }
                }
                V4<T> gq[4], pr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u] && on) {
                        gq[u].load(da + (int64_t)d[u] * ldda + c0);
                        pr[u].load(phi + (int64_t)e[u] * ldphi + c0);
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u] && on) {
                        V4<T> dph;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float dm = __fdiv_rn(gq[u].v[j], cnt[u]);
                            acc[j] = __fadd_rn(acc[j], __fmul_rn(dm, pr[u].v[j]));
                            dph.v[j] = __fmul_rn(dm, xs.v[j]);
                        }
                        if (dphi_out) dph.store(dphi_out + (int64_t)e[u] * lddphi + c0);
                    }
            }
            if (on && rv && dx) {
                V4<T> o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o.v[j] = has_add ? __fadd_rn(acc[j], av.v[j]) : acc[j];
                if (mask_dx) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o.v[j] = xs.v[j] > 0.f ? o.v[j] : 0.f;
                }
                o.store(dx + row * lddx + c0);
            }
        }
    }
}

// the forward in the same form: 64 / G DESTINATION rows per wavefront, each group sums its in-edges in plan order (k_agg_fwd's order: a bit-identical)
template <int G, typename T>
__global__ void __launch_bounds__(256) k_agg_fwd_g(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid,
                                                   int64_t n_dst, const T* __restrict__ x, int64_t ldx, int c_in, const T* __restrict__ phi, int64_t ldphi,
                                                   T* __restrict__ a, int64_t lda, int rows_per_chunk) {
    constexpr int R = 64 / G;
    const int lane = lane_id(), g = lane / G, c0 = 4 * (lane % G);
    const bool on = c0 < c_in;
    const int RW = rows_per_chunk;
    const int64_t nchunks = (n_dst + RW - 1) / RW;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave_id_uniform(); chunk < nchunks; chunk += stride) {
        const int64_t rb = chunk * RW;
        const int nr = (int)(n_dst - rb < RW ? n_dst - rb : RW);
        const int rp = rowptr[rb + (lane < nr ? lane : nr)];
        const int beg0 = rl(rp, 0), ne = rl(rp, nr) - beg0;
        const bool inw = ne <= 64;
        int sv = 0, ev = 0;
        if (inw && lane < ne) {
            sv = src[beg0 + lane];
            ev = eid ? eid[beg0 + lane] : beg0 + lane;
        }
        for (int r0 = 0; r0 < nr; r0 += R) {
            const int r = r0 + g;
            const bool rv = r < nr;
            const int rc = rv ? r : nr - 1;
            const int b = __shfl(rp, rc), deg = rv ? __shfl(rp, rc + 1) - b : 0;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = 0; __any(t0 < deg); t0 += 4) {
                int sj[4], ej[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ok[u] = t0 + u < deg;
                    const int k = b + (ok[u] ? t0 + u : 0);
                    if (inw) {
                        int sl = k - beg0;
                        sl = sl < 0 ? 0 : (sl > 63 ? 63 : sl);
                        sj[u] = __shfl(sv, sl);
                        ej[u] = __shfl(ev, sl);
                    } else {
                        sj[u] = ej[u] = 0;
                        if (ok[u]) {
                            sj[u] = src[k];
                            ej[u] = eid ? eid[k] : k;
                        }
                    }
                }
                V4<T> xr[4], pr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u] && on) {
                        xr[u].load(x + (int64_t)sj[u] * ldx + c0);
                        pr[u].load(phi + (int64_t)ej[u] * ldphi + c0);
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ok[u] && on) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = __fadd_rn(acc[j], __fmul_rn(xr[u].v[j], pr[u].v[j]));
                    }
            }
            if (on && rv) {
                const float cnt = (float)max(deg, 1);
                V4<T> o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o.v[j] = __fdiv_rn(acc[j], cnt);
                o.store(a + (rb + rc) * lda + c0);
            }
        }
    }
}

// the fused-filter forward (Static model, 20 edge attributes) in the lane-group form, for rows of up to 32 channels (G = 8): every lane of a
// group loads its edge's 20 attributes (five 16-byte loads, the same addresses inside a group) and runs the filter's fma chain for its 4 channels in
// k_agg_fwd's order (a bit-identical); one wavefront instruction serves 8 edges instead of one.
template <int G>
__global__ void __launch_bounds__(256) k_agg_fwd_g20(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid,
                                                     int64_t n_dst, const float* __restrict__ x, int64_t ldx, int c_in, const float* __restrict__ ea,
                                                     int64_t lde, const float* __restrict__ We, const float* __restrict__ be, float* __restrict__ a, int64_t lda,
                                                     int rows_per_chunk) {
    constexpr int R = 64 / G, FE = 20;
    const int lane = lane_id(), g = lane / G, c0 = 4 * (lane % G);
    const bool on = c0 < c_in;
    float w[4][FE], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b[j] = 0.f;
#pragma unroll
        for (int f = 0; f < FE; ++f) w[j][f] = 0.f;
        if (on && c0 + j < c_in) {
            b[j] = be[c0 + j];
#pragma unroll
            for (int f = 0; f < FE; ++f) w[j][f] = We[(int64_t)(c0 + j) * FE + f];
        }
    }
    const int RW = rows_per_chunk;
    const int64_t nchunks = (n_dst + RW - 1) / RW;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave_id_uniform(); chunk < nchunks; chunk += stride) {
        const int64_t rb = chunk * RW;
        const int nr = (int)(n_dst - rb < RW ? n_dst - rb : RW);
        const int rp = rowptr[rb + (lane < nr ? lane : nr)];
        const int beg0 = rl(rp, 0), ne = rl(rp, nr) - beg0;
        const bool inw = ne <= 64;
        int sv = 0, ev = 0;
        if (inw && lane < ne) {
            sv = src[beg0 + lane];
            ev = eid ? eid[beg0 + lane] : beg0 + lane;
        }
        for (int r0 = 0; r0 < nr; r0 += R) {
            const int r = r0 + g;
            const bool rv = r < nr;
            const int rc = rv ? r : nr - 1;
            const int bg = __shfl(rp, rc), deg = rv ? __shfl(rp, rc + 1) - bg : 0;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = 0; __any(t0 < deg); t0 += 2) {      // two edges of every row in flight
                int sj[2], ej[2];
                bool ok[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ok[u] = t0 + u < deg;
                    const int k = bg + (ok[u] ? t0 + u : 0);
                    if (inw) {
                        int sl = k - beg0;
                        sl = sl < 0 ? 0 : (sl > 63 ? 63 : sl);
                        sj[u] = __shfl(sv, sl);
                        ej[u] = __shfl(ev, sl);
                    } else {
                        sj[u] = ej[u] = 0;
                        if (ok[u]) {
                            sj[u] = src[k];
                            ej[u] = eid ? eid[k] : k;
                        }
                    }
                }
                V4<float> xr[2];
                float4 A[2][5];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (ok[u] && on) {
                        xr[u].load(x + (int64_t)sj[u] * ldx + c0);
                        const float4* ar = reinterpret_cast<const float4*>(ea + (int64_t)ej[u] * lde);
#pragma unroll
                        for (int q = 0; q < 5; ++q) A[u][q] = ar[q];
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (ok[u] && on) {
                        float p[4] = {b[0], b[1], b[2], b[3]};
#pragma unroll
                        for (int q = 0; q < 5; ++q) {
                            const float av[4] = {A[u][q].x, A[u][q].y, A[u][q].z, A[u][q].w};
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int j = 0; j < 4; ++j) p[j] = __fmaf_rn(w[j][4 * q + i], av[i], p[j]);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = __fadd_rn(acc[j], __fmul_rn(xr[u].v[j], p[j]));
                    }
            }
            if (on && rv) {
                const float cnt = (float)max(deg, 1);
                V4<float> o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o.v[j] = __fdiv_rn(acc[j], cnt);
                o.store(a + (rb + rc) * lda + c0);
            }
        }
    }
}

// =====================================================================================================================
// Fused-filter aggregate, fp32 rows, 20 attributes, the filter product on the fp32 matrix cores (round 6).  phi = We . A + be as
// v_mfma_f32_16x16x4_f32 with the bias as the C input and the attributes ascending is, per (edge, channel), the fmaf chain of k_agg_fwd -- the same bits
// (tests/test_gpu_parity.py: ..._is_the_fmaf_chain, measured on the exact-fp32 fused layers first) -- at the matrix pipe's 32 multiply-adds per clock and SIMD, a rate
// the VALU form's dependent fmaf chains do not reach (nominally the same; fused.hip: 3.9 cycles per VALU instruction measured), with no 20 x 64 weights in registers.  A wavefront step takes 4 EB destination rows; every row gets 4 edge slots (a tetrahedron has 4
// neighbours; a step with a row of more than 4 in-edges takes the per-edge path); lane (n = lane & 15, g = lane >> 4) owns, of row 4 eb + g and of that
// row's 4 neighbour rows, the channels chan(nb, n) = 64 (nb / VW) + VW n + nb % VW (whole cache lines per load instruction), and the C/D layout hands it
// phi of exactly those 4 edges x those channels: the in-order sum over the edges is in-lane.
// =====================================================================================================================
typedef float f32x4m_t __attribute__((ext_vector_type(4)));
template <int NBK>   // blocks of 16 channels: c_in <= 16 NBK
__global__ void __launch_bounds__(256) k_agg_fwd_m(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid,
                                                   int64_t n_dst, const float* __restrict__ x, int64_t ldx, int c_in, const float* __restrict__ ea, int64_t lde,
                                                   const float* __restrict__ We, const float* __restrict__ be, float* __restrict__ a, int64_t lda) {
    constexpr int VW = NBK < 4 ? NBK : 4, NSEG = NBK / VW, EB = 8 / NBK, RS = 4 * EB, FE = 20;
    __shared__ float fwbuf[NBK * 6 * 64];      // [16-channel block][5 k-steps + bias][lane]: B operand We[chan][4 ks + g], C input be[chan]
    const int lane = lane_id(), fn = lane & 15, fg = lane >> 4;
    const int64_t ngrp = (n_dst + RS - 1) / RS, stride = (int64_t)gridDim.x * 4;
    // index pipeline: a step's row pointers are requested two steps ahead, its slots' sources / edge ids one step ahead -- they arrive in the shadow of
    // the step before's row gathers, so a step exposes ONE memory round trip (its gathers), not three
    auto load_rp = [&](int64_t g_, int& rp_, int& nr_) {
        rp_ = 0, nr_ = 0;
        if (g_ < ngrp) {
            const int64_t rb_ = g_ * RS;
            nr_ = (int)(n_dst - rb_ < RS ? n_dst - rb_ : RS);
            rp_ = rowptr[rb_ + (lane < nr_ ? lane : nr_)];
        }
    };
    // lane l <-> slot l of the step: block l >> 4, row 4 (l >> 4) + ((l & 15) >> 2), edge l & 3 of that row
    auto load_slots = [&](int rp_, int nr_, int& sv_, int& ev_, bool& slow_) {
        sv_ = 0, ev_ = -1, slow_ = false;
        if (nr_ > 0) {
            const int rpn = __shfl(rp_, lane + 1 < 64 ? lane + 1 : 63);
            slow_ = __any(lane < nr_ && rpn - rp_ > 4) != 0;
            const int row = 4 * (lane >> 4) + ((lane & 15) >> 2), k_ = lane & 3;
            const int rc = row < nr_ ? row : nr_;
            const int b_ = __shfl(rp_, rc), d_ = row < nr_ ? __shfl(rp_, rc + 1) - b_ : 0;
            if (!slow_ && row < RS && k_ < d_) {
                sv_ = src[b_ + k_];
                ev_ = eid ? eid[b_ + k_] : b_ + k_;
            }
        }
    };
    const int64_t g_first = (int64_t)blockIdx.x * 4 + wave_id_uniform();
    int rp, nr, rp1, nr1, sv, ev;
    bool slow;
    load_rp(g_first, rp, nr);
    load_rp(g_first + stride, rp1, nr1);
    load_slots(rp, nr, sv, ev, slow);
    // the filter operand (filled behind the first index requests: its loads travel with theirs)
    for (int e = threadIdx.x; e < NBK * 6 * 64; e += 256) {
        const int ln = e & 63, nk = e >> 6, nb = nk / 6, ks = nk - 6 * nb;
        const int c = 64 * (nb / VW) + VW * (ln & 15) + nb % VW;
        float v = 0.f;
        if (c < c_in) v = ks < 5 ? We[(int64_t)c * FE + 4 * ks + (ln >> 4)] : be[c];
        fwbuf[e] = v;
    }
    __syncthreads();
    for (int64_t grp = g_first; grp < ngrp; grp += stride) {
        const int64_t rb = grp * RS;
        int sv1, ev1, rp2, nr2;
        bool slow1;
        load_slots(rp1, nr1, sv1, ev1, slow1);
        load_rp(grp + 2 * stride, rp2, nr2);
        if (slow) {
            // ---- per-edge path (never on a Delaunay scene): lane group g takes rows g, g + 4, ...; k_agg_fwd's arithmetic per (row, channel)
            for (int r0 = 0; r0 < nr; r0 += 4) {      // (the shuffles with all lanes in; the groups' edge loops below may differ in length)
                const int r = r0 + fg;
                const bool rv = r < nr;
                const int b_ = __shfl(rp, rv ? r : 0), e_ = rv ? __shfl(rp, rv ? r + 1 : 0) : b_;
                for (int nb = 0; nb < NBK; ++nb) {
                    const int c = 64 * (nb / VW) + VW * fn + nb % VW;
                    if (c >= c_in || !rv) continue;
                    float acc = 0.f;
                    for (int k = b_; k < e_; ++k) {
                        const float* ar = ea + (int64_t)(eid ? eid[k] : k) * lde;
                        float p_ = be[c];
                        for (int f = 0; f < FE; ++f) p_ = __fmaf_rn(We[(int64_t)c * FE + f], ar[f], p_);
                        acc = __fadd_rn(acc, __fmul_rn(x[(int64_t)src[k] * ldx + c], p_));
                    }
                    a[(rb + r) * lda + c] = __fdiv_rn(acc, (float)max(e_ - b_, 1));
                }
            }
        } else {
#pragma unroll
        for (int eb = 0; eb < EB; ++eb) {
            if (4 * eb >= nr) break;
            // A operand: lane (m = slot 16 eb + n, k = g) holds attribute 4 ks + g of its slot's edge (an empty slot: zeros).  (Parking the step's 64 attribute
            // rows in LDS -- five 16-byte loads per slot instead of 4-byte loads per operand element -- measured no faster here and 30 % slower in the backward.)
            const int ae = __shfl(ev, 16 * eb + fn);
            float fa[5];
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) fa[ks] = ae >= 0 ? ea[(int64_t)ae * lde + 4 * ks + fg] : 0.f;
            // the 4 neighbour rows of row 4 eb + g: VW contiguous channels per segment
            const int row = 4 * eb + fg;
            int sj[4];
            bool ok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sj[r] = __shfl(sv, 16 * eb + 4 * fg + r);
                ok[r] = __shfl(ev, 16 * eb + 4 * fg + r) >= 0;
            }
            float xr[4][NBK];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int sg = 0; sg < NSEG; ++sg) {
                    const int cseg = 64 * sg + VW * fn;
                    const float* pp = x + (int64_t)sj[r] * ldx + (cseg < c_in ? cseg : 0);
                    if constexpr (VW == 4) {
                        const f32x4m_t t = *reinterpret_cast<const f32x4m_t*>(pp);
                        xr[r][4 * sg] = t[0], xr[r][4 * sg + 1] = t[1], xr[r][4 * sg + 2] = t[2], xr[r][4 * sg + 3] = t[3];
                    } else {
                        const float2 t = *reinterpret_cast<const float2*>(pp);
                        xr[r][2 * sg] = t.x, xr[r][2 * sg + 1] = t.y;
                    }
                }
            int deg = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) deg += ok[r] ? 1 : 0;
            const float cnt = (float)max(deg, 1);
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) {
                f32x4m_t d[VW];
                float fwv[VW][5];
#pragma unroll
                for (int u = 0; u < VW; ++u) {
                    const float* fwp = fwbuf + (sg * VW + u) * 6 * 64 + lane;
                    const float fb_ = fwp[5 * 64];
                    d[u] = f32x4m_t{fb_, fb_, fb_, fb_};
#pragma unroll
                    for (int ks = 0; ks < 5; ++ks) fwv[u][ks] = fwp[ks * 64];
                }
#pragma unroll
                for (int ks = 0; ks < 5; ++ks)
#pragma unroll
                    for (int u = 0; u < VW; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks], fwv[u][ks], d[u], 0, 0, 0);
                float o[VW];
#pragma unroll
                for (int u = 0; u < VW; ++u) {
                    float acc = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ok[r]) acc = __fadd_rn(acc, __fmul_rn(xr[r][sg * VW + u], d[u][r]));
                    o[u] = __fdiv_rn(acc, cnt);
                }
                const int cseg = 64 * sg + VW * fn;
                if (row < nr && cseg < c_in) {
                    float* op = a + (rb + row) * lda + cseg;
                    if constexpr (VW == 4) *reinterpret_cast<f32x4m_t*>(op) = f32x4m_t{o[0], o[1], o[2], o[3]};
                    else *reinterpret_cast<float2*>(op) = make_float2(o[0], o[1]);
                }
            }
        }
        }
        rp = rp1, nr = nr1, sv = sv1, ev = ev1, slow = slow1, rp1 = rp2, nr1 = nr2;
    }
}

// The backward of the same form (fp32 rows of up to 64 channels; wider rows keep k_agg_bwd_c, whose packed fused multiply-adds run at least at the matrix cores'
// fp32 rate).  A step = 4 EB SOURCE rows x 4 out-edge slots.  Per 16 slots and 16 channels:
//   phi (DX only) as in the forward: 5 matrix instructions, the bits of the VALU chain; lane (n, g) receives phi of row 4 eb + g's 4 out-edges x its channels
//   dm = da[dst(slot)] / in-degree(dst) (a multiplication by the exact reciprocal when every in-degree of the step is a power of two, else the division),
//   dx = sum over the row's slots of dm * phi in slot order (+ the addend row), dphi = dm * x[row] -- in-lane
//   dWe^T [16 channels x 32] += dphi^T . [A | 1 | 0]: k-step r takes slot r of the four rows (k = g): the A operand IS the lane's dphi register r, the B
//        operand attribute n (and 16 + n, 1.0 at column 20) of slot 4 g + r's edge -- 8 matrix instructions, nothing moves between lanes
// The accumulators live for the whole launch; slab per workgroup in k_agg_bwd_c's layout (CPL = 1).  dWe / dbe are sums in ANOTHER order than
// k_agg_bwd_c's (slot-major within a step, steps in the wavefront's walk): fp32-class, deterministic, not the old kernel's bits; dx is (same chains).
template <int NBK, bool DX, bool ADD>
__global__ void __launch_bounds__(256) k_agg_bwd_mm(const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_dst, const int32_t* __restrict__ t_eid,
                                                    int64_t n_src, const int32_t* __restrict__ rowptr_dst, const float* __restrict__ x, int64_t ldx, int c_in,
                                                    const float* __restrict__ ea, int64_t lde, const float* __restrict__ We, const float* __restrict__ be,
                                                    const float* __restrict__ da, int64_t ldda, float* __restrict__ dx, int64_t lddx, float* __restrict__ slabs,
                                                    const float* __restrict__ add, int64_t ldadd, int64_t n_add) {
    static_assert(NBK == 2 || NBK == 4, "rows of up to 64 channels");
    constexpr int VW = NBK, EB = 8 / NBK, RS = 4 * EB, FE = 20;      // one segment: lane n owns channels VW n .. VW n + VW - 1 (block nb <-> channel VW n + nb)
    __shared__ float fwbuf[NBK * 6 * 64];       // the filter operand (DX)
    __shared__ float smem_[4 * 64 * 21];        // the wavefronts' slabs: the per-edge path adds into its wavefront's directly, the accumulators are added at the end
    const int lane = lane_id(), fn = lane & 15, fg = lane >> 4, wv = wave_id_uniform();
    float* const mine = smem_ + wv * (64 * 21);
    for (int i = lane; i < 64 * 21; i += 64) mine[i] = 0.f;
    f32x4m_t accW[NBK][2];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) accW[nb][0] = accW[nb][1] = f32x4m_t{0.f, 0.f, 0.f, 0.f};
    const int64_t ngrp = (n_src + RS - 1) / RS, stride = (int64_t)gridDim.x * 4;
    const bool con = VW * fn < c_in;      // c_in is a multiple of VW (host-checked): a lane's channels are all in or all out
    // index pipeline as in k_agg_fwd_m: row pointers two steps ahead, the slots' destinations / edge ids one step ahead; the destinations' in-degrees are
    // requested with the step's gathers (they are needed behind them)
    auto load_rp = [&](int64_t g_, int& rp_, int& nr_) {
        rp_ = 0, nr_ = 0;
        if (g_ < ngrp) {
            const int64_t rb_ = g_ * RS;
            nr_ = (int)(n_src - rb_ < RS ? n_src - rb_ : RS);
            rp_ = t_rowptr[rb_ + (lane < nr_ ? lane : nr_)];
        }
    };
    // lane l <-> slot l of the step: block l >> 4, row 4 (l >> 4) + ((l & 15) >> 2), out-edge l & 3 of that row
    auto load_slots = [&](int rp_, int nr_, int& dv_, int& ev_, bool& slow_) {
        dv_ = 0, ev_ = -1, slow_ = false;
        if (nr_ > 0) {
            const int rpn = __shfl(rp_, lane + 1 < 64 ? lane + 1 : 63);
            slow_ = __any(lane < nr_ && rpn - rp_ > 4) != 0;
            const int row = 4 * (lane >> 4) + ((lane & 15) >> 2), k_ = lane & 3;
            const int rc = row < nr_ ? row : nr_;
            const int b_ = __shfl(rp_, rc), d_ = row < nr_ ? __shfl(rp_, rc + 1) - b_ : 0;
            if (!slow_ && row < RS && k_ < d_) {
                dv_ = t_dst[b_ + k_];
                ev_ = t_eid[b_ + k_];
            }
        }
    };
    const int64_t g_first = (int64_t)blockIdx.x * 4 + wv;
    int rp, nr, rp1, nr1, dv, ev;
    bool slow;
    load_rp(g_first, rp, nr);
    load_rp(g_first + stride, rp1, nr1);
    load_slots(rp, nr, dv, ev, slow);
    // the filter operand (filled behind the first index requests: its loads travel with theirs)
    if (DX) {
        for (int e = threadIdx.x; e < NBK * 6 * 64; e += 256) {
            const int ln = e & 63, nk = e >> 6, nb = nk / 6, ks = nk - 6 * nb;
            const int c = VW * (ln & 15) + nb;
            float v = 0.f;
            if (c < c_in) v = ks < 5 ? We[(int64_t)c * FE + 4 * ks + (ln >> 4)] : be[c];
            fwbuf[e] = v;
        }
        __syncthreads();
    }
    for (int64_t grp = g_first; grp < ngrp; grp += stride) {
        const int64_t rb = grp * RS;
        int dv1, ev1, rp2, nr2;
        bool slow1;
        load_slots(rp1, nr1, dv1, ev1, slow1);
        load_rp(grp + 2 * stride, rp2, nr2);
        if (slow) {
            // ---- per-edge path (a source row with more than 4 out-edges: never on a Delaunay scene): lane group 0, one row and edge at a time
            for (int r = 0; r < nr; ++r) {
                const int b_ = rl(rp, r), e_ = rl(rp, r + 1);
                if (fg == 0 && con) {
#pragma unroll
                    for (int nb = 0; nb < NBK; ++nb) {
                        const int c = VW * fn + nb;
                        const float xv = x[(rb + r) * ldx + c];
                        float acc = 0.f;
                        for (int k = b_; k < e_; ++k) {
                            const int d_ = t_dst[k];
                            const float* ar = ea + (int64_t)t_eid[k] * lde;
                            const float dm = __fdiv_rn(da[(int64_t)d_ * ldda + c], (float)max(rowptr_dst[d_ + 1] - rowptr_dst[d_], 1));
                            if (DX) {
                                float p_ = be[c];
                                for (int f = 0; f < FE; ++f) p_ = __fmaf_rn(We[(int64_t)c * FE + f], ar[f], p_);
                                acc = __fadd_rn(acc, __fmul_rn(dm, p_));
                            }
                            const float dph = __fmul_rn(dm, xv);
                            mine[c * 21 + 20] += dph;
                            for (int f = 0; f < FE; ++f) mine[c * 21 + f] = __fmaf_rn(dph, ar[f], mine[c * 21 + f]);
                        }
                        if (DX) dx[(rb + r) * lddx + c] = (ADD && rb + r < n_add) ? __fadd_rn(acc, add[(rb + r) * ldadd + c]) : acc;
                    }
                }
            }
        } else {
        int cn0 = 0, cn1 = 1;
        if (ev >= 0) {
            cn0 = rowptr_dst[dv];
            cn1 = rowptr_dst[dv + 1];
        }
        float cf = 1.f, icf = 1.f;
        bool pow2 = true;
#pragma unroll
        for (int eb = 0; eb < EB; ++eb) {
            if (4 * eb >= nr) break;
            const int row = 4 * eb + fg;
            const bool rv = row < nr;
            // phi's A operand: lane (m = slot 16 eb + n, k = g): attribute 4 ks + g of the slot's edge
            float fa[5];
            if (DX) {
                const int ae = __shfl(ev, 16 * eb + fn);
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) fa[ks] = ae >= 0 ? ea[(int64_t)ae * lde + 4 * ks + fg] : 0.f;
            }
            int dj[4], ej[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sl = 16 * eb + 4 * fg + r;
                dj[r] = __shfl(dv, sl);
                ej[r] = __shfl(ev, sl);
            }
            // dWe's B operand: lane (column n, k = g), k-step r: attribute n -- and 16 + n (1.0 at column 20, zeros behind) -- of slot 4 g + r's edge
            float b0[4], b1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* ar = ea + (int64_t)(ej[r] >= 0 ? ej[r] : 0) * lde;
                b0[r] = ej[r] >= 0 ? ar[fn] : 0.f;
                b1[r] = (ej[r] >= 0 && fn < 4) ? ar[16 + fn] : (fn == 4 ? 1.f : 0.f);
            }
            float gq[4][VW], xq[VW], aq[VW];
            auto ldv = [&](float* dst, const float* pp) {
                if constexpr (VW == 4) {
                    const f32x4m_t t = *reinterpret_cast<const f32x4m_t*>(pp);
                    dst[0] = t[0], dst[1] = t[1], dst[2] = t[2], dst[3] = t[3];
                } else {
                    const float2 t = *reinterpret_cast<const float2*>(pp);
                    dst[0] = t.x, dst[1] = t.y;
                }
            };
            const int cs = con ? VW * fn : 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) ldv(gq[r], da + (int64_t)dj[r] * ldda + cs);
            ldv(xq, x + (rb + (rv ? row : 0)) * ldx + cs);
            const bool addrow = ADD && rv && rb + row < n_add;
            if (ADD) {
                if (addrow) ldv(aq, add + (rb + row) * ldadd + cs);
            }
            if (eb == 0) {
                // (behind the first block's gathers in program order: the in-degrees were requested in front of them)
                cf = (float)max(cn1 - cn0, 1);
                // every in-degree of the step a power of two (4 on a Delaunay scene): dm = da * (1 / cnt) is the division's result exactly
                pow2 = __all((__builtin_bit_cast(uint32_t, cf) & 0x007FFFFFu) == 0u) != 0;
                icf = 1.0f / cf;
            }
            float cj[4], ij[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sl = 16 * eb + 4 * fg + r;
                cj[r] = __shfl(cf, sl);
                ij[r] = __shfl(icf, sl);
            }
            f32x4m_t d[VW];
            if (DX) {
                float fwv[VW][5];
#pragma unroll
                for (int u = 0; u < VW; ++u) {
                    const float* fwp = fwbuf + u * 6 * 64 + lane;
                    const float fb_ = fwp[5 * 64];
                    d[u] = f32x4m_t{fb_, fb_, fb_, fb_};
#pragma unroll
                    for (int ks = 0; ks < 5; ++ks) fwv[u][ks] = fwp[ks * 64];
                }
#pragma unroll
                for (int ks = 0; ks < 5; ++ks)
#pragma unroll
                    for (int u = 0; u < VW; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks], fwv[u][ks], d[u], 0, 0, 0);
            }
            float dph[VW][4], o[VW];
#pragma unroll
            for (int u = 0; u < VW; ++u) {
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = ej[r] >= 0 && con && rv;
                    const float dm = ok ? (pow2 ? __fmul_rn(gq[r][u], ij[r]) : __fdiv_rn(gq[r][u], cj[r])) : 0.f;
                    if (DX && ok) acc = __fadd_rn(acc, __fmul_rn(dm, d[u][r]));
                    dph[u][r] = __fmul_rn(dm, xq[u]);
                }
                o[u] = (ADD && addrow) ? __fadd_rn(acc, aq[u]) : acc;
            }
            if (DX && rv && con) {
                float* op = dx + (rb + row) * lddx + cs;
                if constexpr (VW == 4) *reinterpret_cast<f32x4m_t*>(op) = f32x4m_t{o[0], o[1], o[2], o[3]};
                else *reinterpret_cast<float2*>(op) = make_float2(o[0], o[1]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < VW; ++u) {
                    accW[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dph[u][r], b0[r], accW[u][0], 0, 0, 0);
                    accW[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dph[u][r], b1[r], accW[u][1], 0, 0, 0);
                }
        }
        }
        rp = rp1, nr = nr1, dv = dv1, ev = ev1, slow = slow1, rp1 = rp2, nr1 = nr2;
    }
    // ---- the workgroup's slab [64 channels][21]: accW[nb][blk][r] <-> channel VW (4 g + r) + nb, column n + 16 blk; the four wavefronts' sums in wavefront order
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = VW * (4 * fg + r) + nb;      // (every (c, column) of the slab belongs to exactly one lane and register)
            mine[c * 21 + fn] += accW[nb][0][r];
            if (fn < 5) mine[c * 21 + 16 + fn] += accW[nb][1][r];
        }
    __syncthreads();
    constexpr int PER = 64 * 21;
    float* slab = slabs + (int64_t)blockIdx.x * PER;
    for (int i = threadIdx.x; i < PER; i += 256) slab[i] = ((smem_[i] + smem_[PER + i]) + smem_[2 * PER + i]) + smem_[3 * PER + i];
}

inline bool agg_grouped() {   // DGNN_AGG_GROUPED=0: the one-edge-per-instruction kernels for the given-phi backward too
    static const bool on = !(getenv("DGNN_AGG_GROUPED") && getenv("DGNN_AGG_GROUPED")[0] == '0');
    return on;
}
template <typename T>
inline bool rows_of_4(const void* p, int64_t ld) {
    return p == nullptr || (((uintptr_t)p % (4 * sizeof(T))) == 0 && ld % 4 == 0);
}

// rows per chunk: 16 when there are enough rows to give every CU its 16 wavefronts, fewer (down to 4) on the small inner blocks
inline int chunk_rows(int64_t n_rows) {
    int64_t r = dgnn_cdiv(n_rows, (int64_t)DGNN_NUM_CU * 16);
    r = (r + 3) & ~(int64_t)3;
    return (int)(r < 4 ? 4 : (r > CH_ROWS ? CH_ROWS : r));
}
inline bool agg_chunked() {   // DGNN_AGG_CHUNKED=0: the row-at-a-time kernels
    static const bool on = !(getenv("DGNN_AGG_CHUNKED") && getenv("DGNN_AGG_CHUNKED")[0] == '0');
    return on;
}

constexpr int BWD_BLOCKS = 1024;  // 4 blocks (16 waves) per CU: the kernel lives on memory latency; one slab per block

template <int CPL, typename T = float>
bool aligned_for(const void* p, int64_t ld) {
    return CPL == 1 || (((uintptr_t)p % (sizeof(T) * CPL)) == 0 && ld % CPL == 0);
}

template <typename T>
int agg_fwd_t(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const T* x_src, int64_t ldx, int c_in,
              const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be, const T* phi, int64_t ldphi, T* phi_out,
              int64_t ldphi_out, T* a, int64_t lda, hipStream_t stream) {
    DGNN_REQUIRE(n_dst >= 0 && c_in > 0, DGNN_E_INVALID, "aggregate_fwd: bad sizes n_dst=%lld c_in=%d", (long long)n_dst, c_in);
    if (n_dst == 0) return DGNN_OK;
    DGNN_REQUIRE(rowptr && src && x_src && a, DGNN_E_INVALID, "aggregate_fwd: null pointer");
    const bool fused = We != nullptr;
    DGNN_REQUIRE(!fused || (be && edge_attr), DGNN_E_INVALID, "aggregate_fwd: fused mode needs be and edge_attr");
    DGNN_REQUIRE(!fused || f_e == 20 || f_e == 2, DGNN_E_UNSUPPORTED,
                 "aggregate_fwd: fused filter supports f_e in {2,20} (got %d); materialise phi with dgnn_linear_fwd", f_e);
    const bool given = !fused && phi != nullptr;
    if (given && !phi_out && agg_chunked() && agg_grouped() && c_in % 4 == 0 && c_in <= 128 && rows_of_4<T>(x_src, ldx) && rows_of_4<T>(phi, ldphi) &&
        rows_of_4<T>(a, lda)) {
        // lane-group form (see k_agg_bwd_g): 4 channels per lane, 64 / G destination rows per wavefront at once
        const int rw = chunk_rows(n_dst);
        dim3 ggrid((unsigned)dgnn_grid_cap(dgnn_cdiv(dgnn_cdiv(n_dst, rw), 4), 8));
#define LAUNCH_G(GG) hipLaunchKernelGGL((k_agg_fwd_g<GG, T>), ggrid, dim3(256), 0, stream, rowptr, src, eid, n_dst, x_src, ldx, c_in, phi, ldphi, a, lda, rw)
        if (c_in <= 32) LAUNCH_G(8);
        else if (c_in <= 64) LAUNCH_G(16);
        else LAUNCH_G(32);
#undef LAUNCH_G
        return dgnn_check_launch("aggregate_fwd");
    }
    if constexpr (sizeof(T) == 4) {
        // the filter product on the fp32 matrix cores (k_agg_fwd_m; the bits of k_agg_fwd): rows in 8- / 16-byte pieces per lane.  DGNN_AGG_MFMA=0: the VALU forms
        // (a training batch's blocks: 28 channels 35.9 -> 27.5 us, 64 channels 22.7 -> 14.0 us; 128 channels 10.1 -> 10.8 us: those keep k_agg_fwd unless DGNN_AGG_MFMA=2)
        static const bool mfma_on = !(getenv("DGNN_AGG_MFMA") && getenv("DGNN_AGG_MFMA")[0] == '0');
        static const int mfma_max = getenv("DGNN_AGG_MFMA") && getenv("DGNN_AGG_MFMA")[0] == '2' ? 128 : 64;
        const int vw = c_in <= 32 ? 2 : 4;
        if (mfma_on && fused && f_e == 20 && !phi_out && c_in <= mfma_max && c_in % vw == 0 && ldx % vw == 0 && lda % vw == 0 &&
            (((uintptr_t)x_src | (uintptr_t)a) % (4 * vw)) == 0) {
            const int rs = c_in <= 32 ? 16 : (c_in <= 64 ? 8 : 4);
            dim3 mgrid((unsigned)dgnn_grid_cap(dgnn_cdiv(dgnn_cdiv(n_dst, rs), 4), 8));
            if (c_in <= 32)
                hipLaunchKernelGGL((k_agg_fwd_m<2>), mgrid, dim3(256), 0, stream, rowptr, src, eid, n_dst, (const float*)x_src, ldx, c_in, edge_attr, lde, We, be, (float*)a, lda);
            else if (c_in <= 64)
                hipLaunchKernelGGL((k_agg_fwd_m<4>), mgrid, dim3(256), 0, stream, rowptr, src, eid, n_dst, (const float*)x_src, ldx, c_in, edge_attr, lde, We, be, (float*)a, lda);
            else
                hipLaunchKernelGGL((k_agg_fwd_m<8>), mgrid, dim3(256), 0, stream, rowptr, src, eid, n_dst, (const float*)x_src, ldx, c_in, edge_attr, lde, We, be, (float*)a, lda);
            return dgnn_check_launch("aggregate_fwd");
        }
        // (measured on a training batch's blocks: 38 -> 31 us at 28 channels, 19 -> 23 us at 64 -- the five replicated attribute loads per lane
        // and 207 registers eat what the wider instructions save; only the narrow first layer takes this form)
        if (fused && f_e == 20 && !phi_out && agg_chunked() && agg_grouped() && c_in % 4 == 0 && c_in <= 32 && rows_of_4<T>(x_src, ldx) && rows_of_4<T>(a, lda) &&
            ((uintptr_t)edge_attr % 16) == 0 && lde % 4 == 0) {
            const int rw = chunk_rows(n_dst);
            dim3 ggrid((unsigned)dgnn_grid_cap(dgnn_cdiv(dgnn_cdiv(n_dst, rw), 4), 8));
            hipLaunchKernelGGL((k_agg_fwd_g20<8>), ggrid, dim3(256), 0, stream, rowptr, src, eid, n_dst, (const float*)x_src, ldx, c_in, edge_attr, lde, We, be,
                               (float*)a, lda, rw);
            return dgnn_check_launch("aggregate_fwd");
        }
    }
    bool v2 = (c_in % 2 == 0) && (c_in > 64 || sizeof(T) == 2) && aligned_for<2, T>(x_src, ldx) && aligned_for<2, T>(a, lda) &&
              (!given || aligned_for<2, T>(phi, ldphi)) && (!phi_out || aligned_for<2, T>(phi_out, ldphi_out));
    const int cpl = v2 ? 2 : 1;
    const int chunks = (int)dgnn_cdiv(c_in, 64 * cpl);
    dim3 grid(dgnn_grid_cap(dgnn_cdiv(n_dst, 4), 8), chunks), block(256);
#define LAUNCH(CPL, FE)                                                                                               \
    hipLaunchKernelGGL((k_agg_fwd<CPL, FE, T>), grid, block, 0, stream, rowptr, src, eid, n_dst, x_src, ldx, c_in, edge_attr, \
                       lde, We, be, phi, ldphi, phi_out, ldphi_out, a, lda)
    if (fused && f_e == 20) { if (v2) LAUNCH(2, 20); else LAUNCH(1, 20); }
    else if (fused && f_e == 2) { if (v2) LAUNCH(2, 2); else LAUNCH(1, 2); }
    else if (given) { if (v2) LAUNCH(2, 0); else LAUNCH(1, 0); }
    else { if (v2) LAUNCH(2, -1); else LAUNCH(1, -1); }
#undef LAUNCH
    return dgnn_check_launch("aggregate_fwd");
}

template <typename T>
int agg_bwd_t(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src, const int32_t* rowptr_dst,
              const T* x_src, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be,
              const T* phi, int64_t ldphi, const T* da, int64_t ldda, T* dx_src, int64_t lddx, float* dWe, float* dbe, T* dphi_out,
              int64_t lddphi, float* partials, hipStream_t stream, const T* add = nullptr, int64_t ldadd = 0, int64_t n_add = 0,
              SlabReduceDesc* deferred = nullptr, const T* dphi_ext = nullptr, int mask_dx = 0) {
    DGNN_REQUIRE(n_src >= 0 && c_in > 0, DGNN_E_INVALID, "aggregate_bwd: bad sizes");
    DGNN_REQUIRE(!mask_dx || (dx_src && agg_chunked()), DGNN_E_UNSUPPORTED, "aggregate_bwd: the masked dx store lives in the chunked kernels");
    if (n_src == 0) {   // nothing to sum: the parameter gradients are zero (they are written, not accumulated, otherwise)
        if (We && dWe && dbe) {
            (void)hipMemsetAsync(dWe, 0, sizeof(float) * (size_t)c_in * f_e, stream);
            (void)hipMemsetAsync(dbe, 0, sizeof(float) * (size_t)c_in, stream);
        }
        return DGNN_OK;
    }
    DGNN_REQUIRE(t_rowptr && t_dst && t_eid && rowptr_dst && x_src && da, DGNN_E_INVALID, "aggregate_bwd: null pointer");
    const bool fused = We != nullptr;
    DGNN_REQUIRE(!fused || (be && edge_attr && dWe && dbe && partials), DGNN_E_INVALID, "aggregate_bwd: fused mode needs be, edge_attr, dWe, dbe, partials");
    DGNN_REQUIRE(!fused || f_e == 20 || f_e == 2, DGNN_E_UNSUPPORTED, "aggregate_bwd: fused filter supports f_e in {2,20} (got %d)", f_e);
    const bool given = !fused && phi != nullptr;
    bool v2 = (c_in % 2 == 0) && (c_in > 64 || sizeof(T) == 2) && aligned_for<2, T>(x_src, ldx) && aligned_for<2, T>(da, ldda) &&
              (!dx_src || aligned_for<2, T>(dx_src, lddx)) && (!given || aligned_for<2, T>(phi, ldphi)) &&
              (!dphi_out || aligned_for<2, T>(dphi_out, lddphi));
    const int cpl = v2 ? 2 : 1;
    const int chunks = (int)dgnn_cdiv(c_in, 64 * cpl);
    const bool chunked = agg_chunked();
    DGNN_REQUIRE(!add || (chunked && dx_src && ((We != nullptr && f_e == 20) || (We == nullptr && phi != nullptr))), DGNN_E_UNSUPPORTED,
                 "aggregate_bwd: the addend form needs the chunked kernel, dx, and the fused 20-attribute filter or a given phi");
    DGNN_REQUIRE(!dphi_ext || (chunked && We == nullptr && phi != nullptr && dphi_out), DGNN_E_UNSUPPORTED,
                 "aggregate_bwd: dphi_ext needs the chunked kernel in the given-phi form");
    const int rw = chunk_rows(n_src);
    if (given && chunked && agg_grouped() && !dphi_ext && c_in % 4 == 0 && c_in <= 128 && rows_of_4<T>(x_src, ldx) && rows_of_4<T>(phi, ldphi) &&
        rows_of_4<T>(da, ldda) && rows_of_4<T>(dx_src, lddx) && rows_of_4<T>(dphi_out, lddphi) && rows_of_4<T>(add, ldadd)) {
        // lane-group form: 4 channels per lane, 64 / G source rows per wavefront at once
        dim3 ggrid((unsigned)dgnn_grid_cap(dgnn_cdiv(dgnn_cdiv(n_src, rw), 4), 8));
#define LAUNCH_G(GG)                                                                                                                          \
        do { if (add)                                                                                                                         \
            hipLaunchKernelGGL((k_agg_bwd_g<GG, T, true>), ggrid, dim3(256), 0, stream, t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x_src, ldx, c_in, phi, \
                               ldphi, da, ldda, dx_src, lddx, dphi_out, lddphi, rw, add, ldadd, n_add, mask_dx);                             \
        else                                                                                                                                  \
            hipLaunchKernelGGL((k_agg_bwd_g<GG, T, false>), ggrid, dim3(256), 0, stream, t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x_src, ldx, c_in, phi, \
                               ldphi, da, ldda, dx_src, lddx, dphi_out, lddphi, rw, (const T*)nullptr, (int64_t)0, (int64_t)0, mask_dx); } while (0)
        if (c_in <= 32) LAUNCH_G(8);
        else if (c_in <= 64) LAUNCH_G(16);
        else LAUNCH_G(32);
#undef LAUNCH_G
        return dgnn_check_launch("aggregate_bwd");
    }
    if constexpr (sizeof(T) == 4) {
        // rows of up to 32 channels (the first conv layer; DGNN_AGG_MFMA=2: up to 64 -- measured 49 us against k_agg_bwd_c's 46.5 on a training batch's 64-wide
        // layer): the filter's two products on the fp32 matrix cores (k_agg_bwd_mm).  DGNN_AGG_MFMA=0: the VALU form
        static const bool mfma_on = !(getenv("DGNN_AGG_MFMA") && getenv("DGNN_AGG_MFMA")[0] == '0');
        static const int mfma_max = getenv("DGNN_AGG_MFMA") && getenv("DGNN_AGG_MFMA")[0] == '2' ? 64 : 32;
        const int vw = c_in <= 32 ? 2 : 4;
        auto al = [&](const void* p_, int64_t ld_) { return p_ == nullptr || (((uintptr_t)p_ % (4 * vw)) == 0 && ld_ % vw == 0); };
        if (mfma_on && fused && f_e == 20 && c_in <= mfma_max && c_in % vw == 0 && !mask_dx && !dphi_out && al(x_src, ldx) && al(da, ldda) && al(dx_src, lddx) &&
            al(add, ldadd)) {
            const int rs = c_in <= 32 ? 16 : 8;
            const int64_t want_m = dgnn_cdiv(dgnn_cdiv(n_src, rs), 4);
            const int nb_m = (int)(want_m < BWD_BLOCKS ? want_m : BWD_BLOCKS);
#define LAUNCH_M(NBK_, DXV, ADDV)                                                                                                                     \
            hipLaunchKernelGGL((k_agg_bwd_mm<NBK_, DXV, ADDV>), dim3(nb_m), dim3(256), 0, stream, t_rowptr, t_dst, t_eid, n_src, rowptr_dst, (const float*)x_src, \
                               ldx, c_in, edge_attr, lde, We, be, (const float*)da, ldda, (float*)dx_src, lddx, partials, (const float*)add, ldadd, n_add)
            if (c_in <= 32) {
                if (!dx_src) LAUNCH_M(2, false, false);
                else if (add) LAUNCH_M(2, true, true);
                else LAUNCH_M(2, true, false);
            } else {
                if (!dx_src) LAUNCH_M(4, false, false);
                else if (add) LAUNCH_M(4, true, true);
                else LAUNCH_M(4, true, false);
            }
#undef LAUNCH_M
            SlabReduceDesc d;
            d.slabs = partials, d.nblocks = nb_m, d.nchunks = 1, d.per = 64 * 21, d.c_in = c_in, d.fe = 20, d.cpl = 1, d.dWe = dWe, d.dbe = dbe;
            if (deferred)
                *deferred = d;
            else
                hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)slab_reduce_blocks(d)), dim3(16 * RS_SLICES), 0, stream, d);
            return dgnn_check_launch("aggregate_bwd");
        }
    }
    const int64_t want = chunked ? dgnn_cdiv(dgnn_cdiv(n_src, rw), 4) : dgnn_cdiv(n_src, 4);
    const int nblocks = (int)(want < BWD_BLOCKS ? want : BWD_BLOCKS);
    dim3 grid(nblocks, chunks), block(256);
#define LAUNCH(CPL, FE)                                                                                               \
    do { if (chunked && add && (FE == 20 || FE == 0))                                                                 \
        hipLaunchKernelGGL((k_agg_bwd_c<CPL, FE == 0 ? 0 : 20, T, (CPL == 2 && sizeof(T) == 4) ? 4 : 8, true>), grid, block, 0, stream, t_rowptr, t_dst, \
                           t_eid, n_src, rowptr_dst, x_src, ldx, c_in, edge_attr, lde, We, be, phi, ldphi, da, ldda, dx_src, lddx, dphi_out, lddphi, \
                           partials, rw, add, ldadd, n_add, dphi_ext, mask_dx);                                       \
    else if (chunked && FE == 20 && !dx_src && no_dx_form())                                                          \
        hipLaunchKernelGGL((k_agg_bwd_c<CPL, FE == 20 ? 20 : 1, T, (CPL == 2 && sizeof(T) == 4) ? 4 : 8, false, false>), grid, block, 0, stream, t_rowptr, \
                           t_dst, t_eid, n_src, rowptr_dst, x_src, ldx, c_in, edge_attr, lde, We, be, phi, ldphi, da, ldda, dx_src, lddx, dphi_out, \
                           lddphi, partials, rw, (const T*)nullptr, (int64_t)0, (int64_t)0, dphi_ext, mask_dx);       \
    else if (chunked)                                                                                                 \
        hipLaunchKernelGGL((k_agg_bwd_c<CPL, FE, T, (CPL == 2 && sizeof(T) == 4) ? 4 : 8>), grid, block, 0, stream, t_rowptr, t_dst, t_eid, n_src, \
                           rowptr_dst, x_src, ldx, c_in, edge_attr, lde, We, be, phi, ldphi, da, ldda, dx_src, lddx, dphi_out, lddphi, partials, rw, \
                           (const T*)nullptr, (int64_t)0, (int64_t)0, dphi_ext, mask_dx);                             \
    else                                                                                                              \
        hipLaunchKernelGGL((k_agg_bwd<CPL, FE, T>), grid, block, 0, stream, t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x_src, ldx, \
                           c_in, edge_attr, lde, We, be, phi, ldphi, da, ldda, dx_src, lddx, dphi_out, lddphi, partials); } while (0)
    if (fused && f_e == 20) { if (v2) LAUNCH(2, 20); else LAUNCH(1, 20); }
    else if (fused && f_e == 2) { if (v2) LAUNCH(2, 2); else LAUNCH(1, 2); }
    else if (given) { if (v2) LAUNCH(2, 0); else LAUNCH(1, 0); }
    else { if (v2) LAUNCH(2, -1); else LAUNCH(1, -1); }
#undef LAUNCH
    if (fused) {
        SlabReduceDesc d;
        d.slabs = partials, d.nblocks = nblocks, d.nchunks = chunks, d.per = 64 * cpl * (f_e + 1), d.c_in = c_in, d.fe = f_e, d.cpl = cpl, d.dWe = dWe, d.dbe = dbe;
        if (deferred)
            *deferred = d;
        else
            hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)slab_reduce_blocks(d)), dim3(16 * RS_SLICES), 0, stream, d);
    }
    return dgnn_check_launch("aggregate_bwd");
}

}  // namespace

extern "C" int dgnn_sage_aggregate_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst,
                                       const float* x_src, int64_t ldx, int c_in, const float* edge_attr, int64_t lde,
                                       int f_e, const float* We, const float* be, const float* phi, int64_t ldphi,
                                       float* phi_out, int64_t ldphi_out, float* a, int64_t lda, void* stream) {
    return agg_fwd_t<float>(rowptr, src, eid, n_dst, x_src, ldx, c_in, edge_attr, lde, f_e, We, be, phi, ldphi, phi_out, ldphi_out, a, lda,
                            (hipStream_t)stream);
}

extern "C" int dgnn_sage_aggregate_fwd_bf16(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst,
                                            const uint16_t* x_src, int64_t ldx, int c_in, const float* edge_attr, int64_t lde,
                                            int f_e, const float* We, const float* be, const uint16_t* phi, int64_t ldphi,
                                            uint16_t* phi_out, int64_t ldphi_out, uint16_t* a, int64_t lda, void* stream) {
    return agg_fwd_t<uint16_t>(rowptr, src, eid, n_dst, x_src, ldx, c_in, edge_attr, lde, f_e, We, be, phi, ldphi, phi_out, ldphi_out, a, lda,
                               (hipStream_t)stream);
}

extern "C" int64_t dgnn_sage_aggregate_bwd_scratch_elems(int64_t n_src, int c_in, int f_e) {
    (void)n_src;
    if (c_in <= 0 || f_e <= 0) return 1;
    // chunks * 64*cpl >= c_in; cpl in {1,2}: bound with cpl = 1 chunk count * per(cpl=2)
    const int64_t chunks = dgnn_cdiv(c_in, 64);
    return (int64_t)BWD_BLOCKS * chunks * 128 * (f_e + 1);
}

extern "C" int dgnn_sage_aggregate_bwd(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid,
                                       int64_t n_src, const int32_t* rowptr_dst, const float* x_src, int64_t ldx, int c_in,
                                       const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be,
                                       const float* phi, int64_t ldphi, const float* da, int64_t ldda, float* dx_src,
                                       int64_t lddx, float* dWe, float* dbe, float* dphi_out, int64_t lddphi,
                                       float* partials, void* stream) {
    return agg_bwd_t<float>(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x_src, ldx, c_in, edge_attr, lde, f_e, We, be, phi, ldphi, da, ldda,
                            dx_src, lddx, dWe, dbe, dphi_out, lddphi, partials, (hipStream_t)stream);
}

extern "C" int dgnn_sage_aggregate_bwd_bf16(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid,
                                            int64_t n_src, const int32_t* rowptr_dst, const uint16_t* x_src, int64_t ldx, int c_in,
                                            const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be,
                                            const uint16_t* phi, int64_t ldphi, const uint16_t* da, int64_t ldda, uint16_t* dx_src,
                                            int64_t lddx, float* dWe, float* dbe, uint16_t* dphi_out, int64_t lddphi,
                                            float* partials, void* stream) {
    return agg_bwd_t<uint16_t>(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x_src, ldx, c_in, edge_attr, lde, f_e, We, be, phi, ldphi, da, ldda,
                               dx_src, lddx, dWe, dbe, dphi_out, lddphi, partials, (hipStream_t)stream);
}

// dgnn_sage_aggregate_bwd (fused 20-attribute filter, fp32) with dx_src[row] += add[row] for row < n_add folded into the store of dx
extern "C" int dgnn_sage_aggregate_bwd_add(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src,
                                           const int32_t* rowptr_dst, const float* x_src, int64_t ldx, int c_in, const float* edge_attr,
                                           int64_t lde, int f_e, const float* We, const float* be, const float* da, int64_t ldda, float* dx_src,
                                           int64_t lddx, const float* add, int64_t ldadd, int64_t n_add, float* dWe, float* dbe, float* partials,
                                           void* stream) {
    DGNN_REQUIRE(add && n_add >= 0 && n_add <= n_src, DGNN_E_INVALID, "aggregate_bwd_add: bad addend");
    return agg_bwd_t<float>(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x_src, ldx, c_in, edge_attr, lde, f_e, We, be, nullptr, 0, da, ldda, dx_src, lddx,
                            dWe, dbe, nullptr, 0, partials, (hipStream_t)stream, add, ldadd, n_add);
}

int dgnn_sage_aggregate_bwd_add_deferred(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src, const int32_t* rowptr_dst,
                                         const float* x_src, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We,
                                         const float* be, const float* da, int64_t ldda, float* dx_src, int64_t lddx, const float* add, int64_t ldadd,
                                         int64_t n_add, float* dWe, float* dbe, float* partials, void* stream, SlabReduceDesc* desc) {
    DGNN_REQUIRE(desc && We && n_src > 0, DGNN_E_INVALID, "aggregate_bwd_add_deferred: bad arguments");
    return agg_bwd_t<float>(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x_src, ldx, c_in, edge_attr, lde, f_e, We, be, nullptr, 0, da, ldda, dx_src, lddx,
                            dWe, dbe, nullptr, 0, partials, (hipStream_t)stream, add, ldadd, n_add, desc);
}

// given-phi form (Updated variant) with the two additions of a conv layer's backward folded into the stores: dx_src[row] += add[row] for
// row < n_add (add may be NULL), dphi_out[e] = dphi_e + dphi_ext[e] (dphi_ext may be NULL).  bf16: storage 1, fp32: 0.
// library-internal (csrc/train.hip): dgnn_sage_aggregate_bwd_phi_add whose dx store also applies the ReLU mask of the layer below, dx * [x_src > 0]
// (mask_dx != 0; needs the chunked kernels: dgnn_agg_bwd_can_mask()).
bool dgnn_agg_bwd_can_mask() { return agg_chunked(); }
int dgnn_sage_aggregate_bwd_phi_add_masked(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src, const int32_t* rowptr_dst,
                                           const void* x_src, int64_t ldx, int c_in, const void* phi, int64_t ldphi, const void* da, int64_t ldda,
                                           void* dx_src, int64_t lddx, const void* add, int64_t ldadd, int64_t n_add, void* dphi_out, int64_t lddphi,
                                           const void* dphi_ext, int bf16, int mask_dx, void* stream) {
    DGNN_REQUIRE(n_add >= 0 && n_add <= n_src, DGNN_E_INVALID, "aggregate_bwd_phi_add: bad addend");
    if (bf16)
        return agg_bwd_t<uint16_t>(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, (const uint16_t*)x_src, ldx, c_in, nullptr, 0, 0, nullptr, nullptr,
                                   (const uint16_t*)phi, ldphi, (const uint16_t*)da, ldda, (uint16_t*)dx_src, lddx, nullptr, nullptr, (uint16_t*)dphi_out, lddphi,
                                   nullptr, (hipStream_t)stream, (const uint16_t*)add, ldadd, n_add, nullptr, (const uint16_t*)dphi_ext, mask_dx);
    return agg_bwd_t<float>(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, (const float*)x_src, ldx, c_in, nullptr, 0, 0, nullptr, nullptr, (const float*)phi, ldphi,
                            (const float*)da, ldda, (float*)dx_src, lddx, nullptr, nullptr, (float*)dphi_out, lddphi, nullptr, (hipStream_t)stream,
                            (const float*)add, ldadd, n_add, nullptr, (const float*)dphi_ext, mask_dx);
}

extern "C" int dgnn_sage_aggregate_bwd_phi_add(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src, const int32_t* rowptr_dst,
                                               const void* x_src, int64_t ldx, int c_in, const void* phi, int64_t ldphi, const void* da, int64_t ldda,
                                               void* dx_src, int64_t lddx, const void* add, int64_t ldadd, int64_t n_add, void* dphi_out, int64_t lddphi,
                                               const void* dphi_ext, int bf16, void* stream) {
    return dgnn_sage_aggregate_bwd_phi_add_masked(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x_src, ldx, c_in, phi, ldphi, da, ldda, dx_src, lddx, add, ldadd, n_add,
                                                  dphi_out, lddphi, dphi_ext, bf16, 0, stream);
}
