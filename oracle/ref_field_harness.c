/* ref_field_harness.c -- TEST INFRASTRUCTURE.  Thin exported wrappers around the REAL
 * reference's hidden field functions (gf_mul/gf_sqr/gf_isr/..., src/f_field.h:76-84), so that
 * tests/golden/gen_golden.py can capture field-level known answers.  Compiled together with the
 * reference's own sources where they lie (oracle/Makefile target `ref`), output only in
 * oracle/_ref/.  This file contains no reference code. */
#include <string.h>
#include "field.h"

#define EXPORT __attribute__((visibility("default")))

static void in_(gf x, const uint64_t *p) { memcpy(x->limb, p, 64); }
static void out_(uint64_t *p, const gf x) { memcpy(p, x->limb, 64); }

EXPORT void ref_gf_mul(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    gf x, y, z; in_(x, a); in_(y, b); gf_mul(z, x, y); out_(o, z);
}
EXPORT void ref_gf_sqr(uint64_t *o, const uint64_t *a) {
    gf x, z; in_(x, a); gf_sqr(z, x); out_(o, z);
}
EXPORT uint64_t ref_gf_isr(uint64_t *o, const uint64_t *a) {
    gf x, z; in_(x, a); uint64_t m = gf_isr(z, x); out_(o, z); return m;
}
EXPORT void ref_gf_add(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    gf x, y, z; in_(x, a); in_(y, b); gf_add(z, x, y); out_(o, z);
}
EXPORT void ref_gf_sub(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    gf x, y, z; in_(x, a); in_(y, b); gf_sub(z, x, y); out_(o, z);
}
EXPORT void ref_gf_mulw(uint64_t *o, const uint64_t *a, uint32_t w) {
    gf x, z; in_(x, a); gf_mulw_unsigned(z, x, w); out_(o, z);
}
EXPORT void ref_gf_strong_reduce(uint64_t *io) {
    gf x; in_(x, io); gf_strong_reduce(x); out_(io, x);
}
EXPORT void ref_gf_serialize(uint8_t *ser, const uint64_t *a) {
    gf x; in_(x, a); gf_serialize(ser, x);
}
EXPORT uint64_t ref_gf_deserialize(uint64_t *o, const uint8_t *ser) {
    gf x; uint64_t m = gf_deserialize(x, ser, 0); out_(o, x); return m;
}
