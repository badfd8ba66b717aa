// Point arithmetic spread over a quad of lanes, for the serial chains (Horner doublings, the last levels of a reduction tree).
//
// One GPU thread needs ~1100 dependent VALU instructions for an extended-coordinate doubling (4 squarings + 4 multiplications,
// ~2.3 us at one wave per SIMD) -- the host does it in 65 ns, which is why the round loop keeps its Horner chains on the host
// while there are few of them.  Where a launch carries many chains (n_partition = 64: 128 problems per IPP round) they stay on
// the device, and their depth is what the round costs.  The four squarings of a doubling are independent, and so are its four
// products (and the four + four products of an addition): lane q of a quad holds coordinate q of (X, Y, Z, T), every lane runs
// ONE squaring and ONE multiplication per doubling, and the operands move between the lanes with DPP quad_perm moves (VALU rate,
// no LDS).  ~510 instead of ~1100 dependent instructions per doubling, ~680 instead of ~1500 per addition.
//
// Bounds: the operands handed to fd_sq / fd_mul are exactly those of gd_double / gd_add in fe26.hpp (same sums and differences
// of tight values in the same operand positions), so the limb bounds proved there carry over.
#pragma once
#include "fe26.hpp"

namespace rofl {

// value held by lane K of the caller's quad
template <int K> __device__ __forceinline__ fd fq_bcast(const fd &a) {
    fd r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)a.v[i], K * 0x55, 0xf, 0xf, false);
    return r;
}

// coordinate q = lane & 3 of a point: 0 X, 1 Y, 2 Z, 3 T
struct gq { fd v; };

__device__ __forceinline__ gq gq_from_gd(const gd &p, u32 q) {
    gq r; r.v = fd_select(fd_select(p.X, p.Y, q == 1), fd_select(p.Z, p.T, q == 3), q >= 2); return r;
}
// every lane of the quad gets the whole point
__device__ __forceinline__ gd gq_to_gd(const gq &p) {
    gd r; r.X = fq_bcast<0>(p.v); r.Y = fq_bcast<1>(p.v); r.Z = fq_bcast<2>(p.v); r.T = fq_bcast<3>(p.v); return r;
}

// 2 p (all four coordinates: the T product rides on lane 3 at no extra depth)
__device__ __forceinline__ gq gq_double(const gq &p, u32 q) {
    fd X = fq_bcast<0>(p.v), Y = fq_bcast<1>(p.v);
    fd sq = fd_sq(fd_select(p.v, fd_add(X, Y), q == 3));      // XX, YY, ZZ, (X+Y)^2
    fd XX = fq_bcast<0>(sq), YY = fq_bcast<1>(sq), ZZ = fq_bcast<2>(sq), S = fq_bcast<3>(sq);
    fd cY = fd_carry(fd_add(YY, XX));
    fd cZ = fd_carry(fd_sub(YY, XX));
    fd cX = fd_sub(S, cY);
    fd cT = fd_sub(fd_add(ZZ, ZZ), cZ);
    // X3 = cT cX, Y3 = cY cZ, Z3 = cT cZ, T3 = cX cY
    fd f = fd_select(fd_select(cT, cY, q == 1), cX, q == 3);
    fd g = fd_select(fd_select(cZ, cX, q == 0), cY, q == 3);
    gq r; r.v = fd_mul(f, g); return r;
}

// p + r (unified extended addition)
__device__ __forceinline__ gq gq_add(const gq &p, const gq &r, u32 q) {
    fd X1 = fq_bcast<0>(p.v), Y1 = fq_bcast<1>(p.v), X2 = fq_bcast<0>(r.v), Y2 = fq_bcast<1>(r.v);
    // lane 0: (Y1-X1)(Y2-X2)   lane 1: (Y1+X1)(Y2+X2)   lane 2: Z1 Z2   lane 3: T1 T2 (then * 2d)
    fd a = fd_select(fd_select(fd_sub(Y1, X1), fd_add(Y1, X1), q == 1), p.v, q >= 2);
    fd b = fd_select(fd_select(fd_sub(Y2, X2), fd_add(Y2, X2), q == 1), r.v, q >= 2);
    fd m = fd_mul(a, b);
    fd m2 = fd_mul(m, fd_d2());                                 // only lane 3 keeps it
    m = fd_select(m, m2, q == 3);
    fd A = fq_bcast<0>(m), B = fq_bcast<1>(m), ZZ = fq_bcast<2>(m), C = fq_bcast<3>(m);
    fd D = fd_add(ZZ, ZZ);
    fd E = fd_sub(B, A), H = fd_add(B, A), F = fd_sub(D, C), G = fd_add(D, C);
    // X3 = F E, Y3 = G H, Z3 = F G, T3 = E H
    fd f = fd_select(fd_select(F, G, q == 1), E, q == 3);
    fd g = fd_select(fd_select(fd_select(E, H, q == 1), G, q == 2), H, q == 3);
    gq o; o.v = fd_mul(f, g); return o;
}

}  // namespace rofl
