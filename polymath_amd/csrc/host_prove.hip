// pm_host_prove: the reference's create_proof_with_assignment glue (prover.rs:66-237) inside the library,
// for hosts that do not bring their own Transcript: the C++ host mirror (polymath_amd/host/polymath.hpp --
// transcripts of src/transcript/*.rs, challenge arithmetic of common.rs:21-98, ark wire format) driven on the
// caller's context.  The three phases stay the boundary; this is their caller, compiled once.
#include "internal.h"
#include "../host/polymath.hpp"

namespace {

template <class C, class T>
int host_prove_impl(pm_ctx *ctx, const pm_pk *pk, const uint64_t *instance_host, const uint64_t *x, const uint64_t *w, int on_device,
                    const uint64_t *r_a, pm_combine_fn combine, void *user, uint8_t *proof_bytes, size_t cap, size_t *proof_len) {
    typedef pmhost::FrOps<C> F;
    typedef typename F::Fr Fr;
    pmhost::Context view(ctx, pmhost::Context::Borrow{});
    pmhost::ProvingKey<C> key;
    key.h = const_cast<pm_pk *>(pk);
    key.n = pk->n; key.m0 = pk->m0; key.sigma = pk->sigma;
    memcpy(key.omega.l, pk->omega, 32);
    std::vector<Fr> instance(pk->m0);
    memcpy((void *)instance.data(), instance_host, pk->m0 * sizeof(Fr));
    Fr ra[2];
    memcpy(ra, r_a, sizeof(ra));
    int status = PM_OK;
    pm::timing_reset(ctx);
    ctx->keep_timings = true;      // pm_last_timings then covers the whole proof
    try {
        pmhost::Polymath<C, T> pm(view);
        typename pmhost::Polymath<C, T>::Combine cb = nullptr;
        if (!combine && pk->shard_count != 1 && ctx->comm) {   // the context's own communicator: all-gather + pm_g1_sum, no callback
            combine = [](void *user, int count, uint64_t *xy, int *inf) -> int { return pm_comm_combine_points((pm_comm *)user, C::ID, count, xy, inf); };
            user = ctx->comm;
        }
        if (combine)
            cb = [&](pmhost::G1Point<C> *pts, int count) -> int {
                uint64_t xy[2][sizeof(pm::Affine<C>) / 8];
                int inf[2];
                for (int i = 0; i < count; ++i) { memcpy(xy[i], &pts[i].p, sizeof(pm::Affine<C>)); inf[i] = pts[i].inf ? 1 : 0; }
                const int rc = combine(user, count, &xy[0][0], inf);
                for (int i = 0; i < count; ++i) { memcpy(&pts[i].p, xy[i], sizeof(pm::Affine<C>)); pts[i].inf = inf[i] != 0; }
                return rc;
            };
        pmhost::Proof<C> proof = pm.prove_raw(key, instance, x, w, on_device != 0, ra, cb);
        pmhost::Bytes b = proof.to_bytes();
        if (proof_len) *proof_len = b.size();
        if (b.size() > cap) status = PM_ERR_INVALID_ARG;
        else memcpy(proof_bytes, b.data(), b.size());
    } catch (const pmhost::PolymathError &e) {
        status = e.status ? e.status : PM_ERR_STATE;
    } catch (const std::exception &e) {
        ctx->err = e.what();
        status = PM_ERR_STATE;
    }
    ctx->keep_timings = false;
    key.h = nullptr;   // borrowed: the destructor must not free the caller's key
    return status;
}

template <class C>
int host_prove_curve(pm_ctx *ctx, const pm_pk *pk, int transcript, const uint64_t *ih, const uint64_t *x, const uint64_t *w, int dev,
                     const uint64_t *r_a, pm_combine_fn cf, void *user, uint8_t *out, size_t cap, size_t *len) {
    switch (transcript) {
        case PM_TRANSCRIPT_MERLIN: return host_prove_impl<C, pmhost::MerlinFieldTranscript<C>>(ctx, pk, ih, x, w, dev, r_a, cf, user, out, cap, len);
        case PM_TRANSCRIPT_KECCAK256: return host_prove_impl<C, pmhost::Keccak256Transcript<C>>(ctx, pk, ih, x, w, dev, r_a, cf, user, out, cap, len);
        case PM_TRANSCRIPT_BLAKE3: return host_prove_impl<C, pmhost::Blake3Transcript<C>>(ctx, pk, ih, x, w, dev, r_a, cf, user, out, cap, len);
        default: return PM_ERR_INVALID_ARG;
    }
}

}  // namespace

extern "C" int pm_host_prove_sharded(pm_ctx *ctx, const pm_pk *pk, int transcript, const uint64_t *instance_host, const uint64_t *x,
                                     const uint64_t *w, int assignment_on_device, const uint64_t *r_a, pm_combine_fn combine, void *user,
                                     uint8_t *proof_bytes, size_t capacity, size_t *proof_len) {
    if (!ctx || !pk || !instance_host || !x || !r_a || !proof_bytes || (pk->mw && !w)) return PM_ERR_INVALID_ARG;
    if (pk->device != ctx->device) return PM_ERR_INVALID_ARG;
    if (pk->shard_count != 1 && !combine && !ctx->comm) return PM_ERR_INVALID_ARG;   // a shard's points are partial sums: somebody has to add them
    if (hipSetDevice(ctx->device) != hipSuccess) return PM_ERR_HIP;
    return pk->curve == PM_BLS12_381
               ? host_prove_curve<pm::BlsCurve>(ctx, pk, transcript, instance_host, x, w, assignment_on_device, r_a, combine, user, proof_bytes, capacity, proof_len)
               : host_prove_curve<pm::BnCurve>(ctx, pk, transcript, instance_host, x, w, assignment_on_device, r_a, combine, user, proof_bytes, capacity, proof_len);
}

extern "C" int pm_host_prove(pm_ctx *ctx, const pm_pk *pk, int transcript, const uint64_t *instance_host, const uint64_t *x,
                             const uint64_t *w, int assignment_on_device, const uint64_t *r_a, uint8_t *proof_bytes, size_t capacity,
                             size_t *proof_len) {
    if (pk && pk->shard_count != 1) return PM_ERR_INVALID_ARG;
    return pm_host_prove_sharded(ctx, pk, transcript, instance_host, x, w, assignment_on_device, r_a, nullptr, nullptr, proof_bytes, capacity,
                                 proof_len);
}
