// coalesce.hpp -- lane-contiguous access to arrays of 32-byte field elements for the HBM-bound elementwise kernels.
//
// fp_load (field.hpp) makes every lane read its own element with two 16-byte loads at a 32-byte lane stride: each wave
// instruction touches 2 KiB of address space for 1 KiB of data, every 128-byte line is requested by two instructions.
// Round 1 measured 4.5-5.4 TB/s for the elementwise family against the 6.3 TB/s a 16-byte-per-lane copy reaches.
// Here a wave walks 128-element blocks with FOUR instructions that each read 1 KiB of consecutive bytes (lane L reads
// 16-byte chunk 64 j + L of the block, j = 0..3): a lane ends up with the SAME half (L & 1) of four elements, and one
// swap with its neighbour (a DPP quad_perm, no LDS) turns that into two whole elements per lane:
//      even lane L : elements  L/2        and  L/2 + 64         of the block
//      odd  lane L : elements  L/2 + 32   and  L/2 + 96
// Stores run the same exchange backwards, so every lane writes back exactly the chunks it read: in-place operation stays
// safe.  Element indices go through a caller-supplied map (rotations wrap per chunk, tails are masked per element).
#pragma once
#include "field.hpp"

namespace h2 {

__device__ __forceinline__ uint32_t dpp_swap_pair(uint32_t v) {
    // quad_perm [1, 0, 3, 2]: exchange with the neighbouring lane
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
}
__device__ __forceinline__ uint4 dpp_swap_pair(uint4 v) {
    return make_uint4(dpp_swap_pair(v.x), dpp_swap_pair(v.y), dpp_swap_pair(v.z), dpp_swap_pair(v.w));
}
__device__ __forceinline__ uint4 sel4(bool c, uint4 a, uint4 b) {  // c ? a : b
    return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
template <class P>
__device__ __forceinline__ Fp<P> fp_from_halves(uint4 lo, uint4 hi) {
    Fp<P> r;
    r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
    r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
    return r;
}

static constexpr size_t CO_BLOCK = 128;  // elements a wave handles per step (two per lane)
static constexpr size_t CO_NONE = ~(size_t)0;

// logical index (inside the array) of this lane's element `which` (0 / 1) of the block starting at `blk`
__device__ __forceinline__ size_t co_element(size_t blk, uint32_t lane, int which) {
    return blk + (lane >> 1) + ((lane & 1) ? 32 : 0) + (which ? 64 : 0);
}

// Loads this lane's two elements of the block at `blk`.  `phys(e)` maps a logical element index to the index inside `p`
// (rotation) or CO_NONE when e is out of range; out-of-range elements read as zero.
template <class P, class Map>
__device__ __forceinline__ void co_load2(const Fp<P>* p, size_t blk, uint32_t lane, Map phys, Fp<P>& a, Fp<P>& b) {
    const bool odd = lane & 1;
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const size_t e = phys(blk + 32 * j + (lane >> 1));
        v[j] = make_uint4(0, 0, 0, 0);
        if (e != CO_NONE) v[j] = *(reinterpret_cast<const uint4*>(p + e) + (odd ? 1 : 0));
    }
    const uint4 r0 = dpp_swap_pair(sel4(odd, v[0], v[1]));
    const uint4 r1 = dpp_swap_pair(sel4(odd, v[2], v[3]));
    a = fp_from_halves<P>(sel4(odd, r0, v[0]), sel4(odd, v[1], r0));
    b = fp_from_halves<P>(sel4(odd, r1, v[2]), sel4(odd, v[3], r1));
}

// Stores this lane's two results; `valid(e)` masks the tail.
template <class P, class Valid>
__device__ __forceinline__ void co_store2(Fp<P>* p, size_t blk, uint32_t lane, Valid valid, const Fp<P>& a, const Fp<P>& b) {
    const bool odd = lane & 1;
    const uint4 alo = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]), ahi = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
    const uint4 blo = make_uint4(b.l[0], b.l[1], b.l[2], b.l[3]), bhi = make_uint4(b.l[4], b.l[5], b.l[6], b.l[7]);
    const uint4 r0 = dpp_swap_pair(sel4(odd, alo, ahi));
    const uint4 r1 = dpp_swap_pair(sel4(odd, blo, bhi));
    uint4 w[4];
    w[0] = sel4(odd, r0, alo);
    w[1] = sel4(odd, ahi, r0);
    w[2] = sel4(odd, r1, blo);
    w[3] = sel4(odd, bhi, r1);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const size_t e = blk + 32 * j + (lane >> 1);
        if (valid(e)) *(reinterpret_cast<uint4*>(p + e) + (odd ? 1 : 0)) = w[j];
    }
}

}  // namespace h2
