// pm_host_prove: the reference's create_proof_with_assignment glue (prover.rs:66-237) inside the library,
// for hosts that do not bring their own Transcript: the C++ host mirror (polymath_amd/host/polymath.hpp --
// transcripts of src/transcript/*.rs, challenge arithmetic of common.rs:21-98, ark wire format) driven on the
// caller's context.  The three phases stay the boundary; this is their caller, compiled once.
#include "internal.h"
#include "../host/polymath.hpp"
#include "../host/wire.hpp"

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
    ctx->lazy_timings = true;      // ... and reads the stage timers when asked, not between the phases
    try {
        pmhost::Polymath<C, T> pm(view);
        typename pmhost::Polymath<C, T>::Combine cb = nullptr;
        if (pk->layout == PM_SHARD_VECTOR) {
            combine = nullptr;     // the phases of a PM_SHARD_VECTOR key return points already summed over the ranks (one exchange per phase)
        } else if (!combine && pk->shard_count != 1 && ctx->comm) {   // the context's own communicator: all-gather + pm_g1_sum, no callback
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
    ctx->lazy_timings = false;
    if (ctx->aux) ctx->aux->lazy_timings = false;
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

// ---- verify (lib.rs:80-90 -> verifier.rs:19-62) and the verifying key (generator.rs:139-157): host code, no GPU --------------
namespace {

template <class C>
int make_vk_impl(uint64_t n, uint64_t m0, uint64_t sigma, const uint64_t *omega, const uint64_t *x_trap, const uint64_t *z_trap, uint8_t *out,
                 size_t cap, size_t *len) {
    typedef typename pmhost::FrOps<C>::Fr Fr;
    pmhost::ProvingKey<C> shape;           // carries n / m0 / sigma / omega only (no device handle)
    shape.n = n; shape.m0 = m0; shape.sigma = sigma;
    memcpy(shape.omega.l, omega, 32);
    Fr x, z;
    memcpy(x.l, x_trap, 32);
    memcpy(z.l, z_trap, 32);
    const pmhost::VerifyingKeyT<C> vk = pmhost::Polymath<C, pmhost::MerlinFieldTranscript<C>>::make_vk(shape, x, z);
    pmhost::Bytes b;
    pmhost::ser_vk_c<C>(vk, b);
    if (len) *len = b.size();
    if (b.size() > cap) return PM_ERR_INVALID_ARG;
    memcpy(out, b.data(), b.size());
    return PM_OK;
}

template <class C, class T>
int verify_impl(const uint8_t *vk_bytes, size_t vk_len, const uint64_t *inputs, size_t n_inputs, const uint8_t *proof_bytes, size_t proof_len,
                int *accepted) {
    typedef typename pmhost::FrOps<C>::Fr Fr;
    pmhost::Reader rd(vk_bytes, vk_len);
    const pmhost::VerifyingKeyT<C> vk = pmhost::read_vk_c<C>(rd);
    if (rd.off != vk_len) return PM_ERR_INVALID_ARG;
    const pmhost::Proof<C> proof = pmhost::read_proof<C>(proof_bytes, proof_len);
    std::vector<Fr> pub(n_inputs);
    if (n_inputs) memcpy((void *)pub.data(), inputs, n_inputs * sizeof(Fr));
    *accepted = pmhost::Polymath<C, T>::verify(vk, pub, proof) ? 1 : 0;
    return PM_OK;
}

template <class C>
int verify_curve(int transcript, const uint8_t *vk, size_t vk_len, const uint64_t *in, size_t n_in, const uint8_t *pr, size_t pr_len, int *acc) {
    switch (transcript) {
        case PM_TRANSCRIPT_MERLIN: return verify_impl<C, pmhost::MerlinFieldTranscript<C>>(vk, vk_len, in, n_in, pr, pr_len, acc);
        case PM_TRANSCRIPT_KECCAK256: return verify_impl<C, pmhost::Keccak256Transcript<C>>(vk, vk_len, in, n_in, pr, pr_len, acc);
        case PM_TRANSCRIPT_BLAKE3: return verify_impl<C, pmhost::Blake3Transcript<C>>(vk, vk_len, in, n_in, pr, pr_len, acc);
        default: return PM_ERR_INVALID_ARG;
    }
}

}  // namespace

extern "C" int pm_host_make_vk(int curve, uint64_t n, uint64_t m0, uint64_t sigma, const uint64_t *omega, const uint64_t *x_trapdoor,
                               const uint64_t *z_trapdoor, uint8_t *vk_bytes, size_t capacity, size_t *vk_len) {
    if (!omega || !x_trapdoor || !z_trapdoor || !vk_bytes) return PM_ERR_INVALID_ARG;
    try {
        if (curve == PM_BLS12_381) return make_vk_impl<pm::BlsCurve>(n, m0, sigma, omega, x_trapdoor, z_trapdoor, vk_bytes, capacity, vk_len);
        if (curve == PM_BN254) return make_vk_impl<pm::BnCurve>(n, m0, sigma, omega, x_trapdoor, z_trapdoor, vk_bytes, capacity, vk_len);
    } catch (const std::exception &) {
        return PM_ERR_STATE;
    }
    return PM_ERR_INVALID_ARG;
}

extern "C" int pm_host_verify(int curve, int transcript, const uint8_t *vk_bytes, size_t vk_len, const uint64_t *public_inputs, size_t n_inputs,
                              const uint8_t *proof_bytes, size_t proof_len, int *accepted) {
    if (!vk_bytes || !proof_bytes || !accepted || (n_inputs && !public_inputs)) return PM_ERR_INVALID_ARG;
    *accepted = 0;
    try {
        if (curve == PM_BLS12_381) return verify_curve<pm::BlsCurve>(transcript, vk_bytes, vk_len, public_inputs, n_inputs, proof_bytes, proof_len, accepted);
        if (curve == PM_BN254) return verify_curve<pm::BnCurve>(transcript, vk_bytes, vk_len, public_inputs, n_inputs, proof_bytes, proof_len, accepted);
    } catch (const std::exception &) {      // malformed vk / proof bytes (off-curve points, non-canonical scalars, truncation)
        return PM_ERR_INVALID_ARG;
    }
    return PM_ERR_INVALID_ARG;
}
