// abi.hpp -- conversion between the reference's C ABI layouts and lane registers.
//   gf_448_s   8 x u64 limbs of 56 bits   (src/public_include/goldilocks/point_448.h:33-35)
//   point_s    {x,y,z,t} = 32 x u64        (point_448.h:66-70)
//   scalar_s   7 x u64 little-endian       (point_448.h:82-86)
//   niels      {a,b,c} canonical limbs     (src/goldilocks.c:55-59; c = 2 d' T, i.e. -cn)
#pragma once
#include "point.hpp"
#include "sc14.hpp"

namespace gd {

GD_FN pt pt_from_abi(const uint64_t *l) {  // 32 limbs
    pt p;
    p.x = fe_weak(fe_from_limbs56(l));
    p.y = fe_weak(fe_from_limbs56(l + 8));
    p.z = fe_weak(fe_from_limbs56(l + 16));
    p.t = fe_weak(fe_from_limbs56(l + 24));
    return p;
}
GD_FN void pt_to_abi(uint64_t *l, const pt &p) {
    fe_to_limbs56(l, p.x);
    fe_to_limbs56(l + 8, p.y);
    fe_to_limbs56(l + 16, p.z);
    fe_to_limbs56(l + 24, p.t);
}
GD_FN sc sc_from_abi(const uint64_t *l) {  // 7 limbs
    sc s;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        s.w[2 * i] = (uint32_t)l[i];
        s.w[2 * i + 1] = (uint32_t)(l[i] >> 32);
    }
    return s;
}
GD_FN void sc_to_abi(uint64_t *l, const sc &s) {
#pragma unroll
    for (int i = 0; i < 7; i++) l[i] = (uint64_t)s.w[2 * i] | ((uint64_t)s.w[2 * i + 1] << 32);
}
// reference-format affine niels (24 limbs) -> ours (negates c once)
GD_FN niels niels_from_abi(const uint64_t *l) {
    niels e;
    e.a = fe_weak(fe_from_limbs56(l));
    e.b = fe_weak(fe_from_limbs56(l + 8));
    e.cn = fe_weak(fe_neg(fe_from_limbs56(l + 16)));
    return e;
}

}  // namespace gd
