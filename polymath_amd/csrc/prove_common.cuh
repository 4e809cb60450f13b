// Pieces shared by the two provers (prove.hip: whole vectors on one GPU; prove_sharded.hip: PM_SHARD_VECTOR).
#pragma once
#include <cstring>

#include "internal.h"

namespace pm {

// --------------------------------------------------------------------------- witness map
// rows 2m0+r and 2m0+nr+r of (U z, W z) and y_{m0+r} = ((A-B) xw)_r^2  (prover.rs:279-302,
// common.rs:138-207).  CSR values are Montgomery Fr.
template <class P>
__device__ __forceinline__ Fp<P> csr_row_dot(const uint64_t *rowptr, const uint32_t *col, const uint64_t *val,
                                             const Fp<P> *z, uint64_t r) {
    Fp<P> acc = Fp<P>::zero();
    for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        Fp<P> v = *(const Fp<P> *)(val + 4 * k);
        acc = add<P>(acc, mul<P>(v, z[col[k]]));
    }
    return acc;
}

struct CsrDev {
    const uint64_t *rowptr;
    const uint32_t *col;
    const uint64_t *val;
};

struct NumParams {
    uint64_t n, sigma, len;
};

template <class P>
struct NumConsts {
    Fp<P> x2, r0, r1, x2r0, x2r1, b2[3], two_x2_r0, two_x2_r1, minus_const;
};

// ------------------------------------------------------------------------------- helpers
template <class P>
static inline Fp<P> load_fr(const uint64_t *p) {
    Fp<P> r;
    memcpy(r.l, p, sizeof(r.l));
    return r;
}
template <class C>
static inline void store_affine_host(const Affine<C> &a, int inf, uint64_t *xy, int *out_inf) {
    if (inf) memset(xy, 0, sizeof(Affine<C>));
    else memcpy(xy, &a, sizeof(Affine<C>));
    *out_inf = inf;
}

static inline unsigned nblk(uint64_t n, unsigned b = 256) { return (unsigned)((n + b - 1) / b); }


// the constants of the numerator (prover.rs:145-197) from the challenges and r_a
template <class P>
static inline NumConsts<P> make_num_consts(const Fp<P> &x2, const Fp<P> rah[2], const Fp<P> &a_at, const Fp<P> &c_at) {
    NumConsts<P> nc;
    nc.x2 = x2;
    nc.r0 = rah[0];
    nc.r1 = rah[1];
    nc.x2r0 = mul<P>(x2, rah[0]);
    nc.x2r1 = mul<P>(x2, rah[1]);
    nc.b2[0] = add<P>(rah[0], mul<P>(x2, sqr<P>(rah[0])));
    nc.b2[1] = add<P>(rah[1], mul<P>(x2, dbl<P>(mul<P>(rah[0], rah[1]))));
    nc.b2[2] = mul<P>(x2, sqr<P>(rah[1]));
    nc.two_x2_r0 = dbl<P>(nc.x2r0);
    nc.two_x2_r1 = dbl<P>(nc.x2r1);
    nc.minus_const = neg<P>(add<P>(a_at, mul<P>(x2, c_at)));
    return nc;
}

}  // namespace pm
