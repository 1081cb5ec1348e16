// gmmvi_train_iter_samtron: the whole SAMTRON iteration (GMMVI.train_iter, optimization/gmmvi.py:146-174, with a
// component-based sample selector at reuse ratio 0) issued from ONE host call.  It is exactly the composition of the
// public entry points of this library in the order the Python modules call them -- no extra arithmetic lives here --
// (the element-wise bookkeeping between them -- DB mapping offset, model snapshot into the DB, the two stepsize rules,
// the weight-history column -- is folded into one prep kernel / the weight-update kernel, same arithmetic per element)
// so that the fast path and the modular plug-in path produce identical results (tests/test_hip_fused.py).  Purpose:
// the modular path costs ~20 Python->C transitions per iteration (~340 us of host time, more than the kernels take).
#include "common.h"
#include "stepsize_rules.h"
#include "iter_prep.h"
#include "stein_finalize.h"

namespace {
struct Arena {
    float *ld, *lq, *qgrad, *bg, *H, *g, *E;
    int32_t *mapping, *success;
};

}  // namespace

static int arena_reserve(gmmvi_ctx* ctx, size_t floats) {
    if (floats * sizeof(float) <= ctx->arena_bytes) return GMMVI_OK;
    GMMVI_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->arena) GMMVI_HIP_CHECK(ctx, hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_bytes = 0;
    size_t want = floats * sizeof(float) * 3 / 2;
    GMMVI_HIP_CHECK(ctx, hipMalloc(&ctx->arena, want));
    ctx->arena_bytes = want;
    return GMMVI_OK;
}

#define GMMVI_TRY(call)                \
    do {                               \
        int rc__ = (call);             \
        if (rc__ != GMMVI_OK) return rc__; \
    } while (0)

static int train_iter_samtron_body(gmmvi_ctx* ctx, const gmmvi_samtron_plan* p);

extern "C" int gmmvi_train_iter_samtron(gmmvi_ctx* ctx, const gmmvi_samtron_plan* p) {
    GMMVI_ARG_CHECK(ctx, ctx != nullptr);
    // a deferred merge points into defer_ws and the arena: none may survive this call on an error exit (a later public call
    // on the context would carry or flush it into memory arena_reserve may have freed), none is inherited at entry
    int rc = gmmvi_flush_pending_combine(ctx);
    if (rc == GMMVI_OK) rc = train_iter_samtron_body(ctx, p);
    if (rc != GMMVI_OK) {
        ctx->pending = CombineJob();
        ctx->riders.prep_blocks = ctx->riders.sample_blocks = 0;
        ctx->defer_combine = false;
        ctx->prof_tag = nullptr;
    }
    return rc;
}

static int train_iter_samtron_body(gmmvi_ctx* ctx, const gmmvi_samtron_plan* p) {
    GMMVI_ARG_CHECK(ctx, p != nullptr);
    const int K = p->K, D = p->D, N = p->N;
    GMMVI_ARG_CHECK(ctx, K >= 1 && D >= 1 && D < GMMVI_MAX_DIM && N >= 1);
    GMMVI_ARG_CHECK(ctx, p->means && p->chols && p->logw && p->packed && p->packed_new && p->stepsizes && p->last_eta &&
                             p->l2 && p->num_updates && p->offsets && p->bg_logw && p->db_samples && p->db_tlp &&
                             p->db_tgrad && p->db_mapping && p->reward_next && p->wstate);
    GMMVI_ARG_CHECK(ctx, p->n_old >= 0 && p->max_per_component >= 0 && p->phase >= 0 && p->phase <= 2);
    GMMVI_ARG_CHECK(ctx, p->n_old == 0 || (p->bg_packed != nullptr && p->bg_K >= 1 && p->bg_old != nullptr && p->bg_logw_new != nullptr));
    const int n_old = p->n_old;
    const int Na = n_old + N;                          // active samples: the reused ones followed by the new ones
    const size_t KN = (size_t)K * Na, ND = (size_t)Na * D;
    const size_t floats = KN + 4 * (size_t)Na + ND + (size_t)K * D * D + (size_t)K * D + 3 * (size_t)K + Na + 64;
    GMMVI_TRY(arena_reserve(ctx, floats));
    Arena a;
    float* base = (float*)ctx->arena;
    a.ld = base; base += KN;
    a.lq = base; base += Na;
    a.bg = base; base += Na;
    a.qgrad = base; base += ND;
    a.H = base; base += (size_t)K * D * D;
    a.g = base; base += (size_t)K * D;
    a.E = base; base += K;
    a.success = (int32_t*)base; base += K;
    a.mapping = (int32_t*)base; base += Na;
    float* bg_a = base; base += Na;                    // reused samples: the two halves of the background density
    float* bg_b = base; base += Na;

    const int phase = p->phase;        // 0: everything; 1: through the component update; 2: the weight update
    float* x = p->db_samples;          // the new samples are written straight into the database
    float* xa = x - (size_t)n_old * D; // the active samples: the n_old database rows in front of them and the new ones
    const float* tlp_a = p->db_tlp - n_old;
    const float* tgrad_a = p->db_tgrad - (size_t)n_old * D;
    if (phase != 2) {
    // ---- sample selection: draw, evaluate the target, append to the DB (sample_selector.py:160-219) --------------------
    // the element-wise bookkeeping (model snapshot into the DB, the two stepsize rules) rides in extra blocks of the
    // sampling launch, and the sampling blocks write the DB mapping (component index + base) directly
    PrepArgs prep_later{};
    bool prep_pending = false;
    {
        PrepArgs q{};
        if (p->db_means && p->db_chols && p->db_packed) {
            q.cdst[0] = (uint32_t*)p->db_means; q.csrc[0] = (const uint32_t*)p->means; q.cwords[0] = (size_t)K * D;
            q.cdst[1] = (uint32_t*)p->db_chols; q.csrc[1] = (const uint32_t*)p->chols; q.cwords[1] = (size_t)K * D * D;
            q.cdst[2] = (uint32_t*)p->db_packed; q.csrc[2] = (const uint32_t*)p->packed;
            q.cwords[2] = (size_t)K * gmmvi_packed_stride(D);
        }
        q.K = K; q.cs_mode = p->component_stepsize_mode; q.stepsizes = p->stepsizes;
        q.reward_prev = p->reward_prev; q.reward_last = p->reward_last;
        q.cs_min = p->cs_min; q.cs_max = p->cs_max; q.cs_inc = p->cs_inc; q.cs_dec = p->cs_dec;
        q.ws_mode = p->weight_stepsize_mode; q.logw = p->logw; q.wstate = p->wstate;
        q.ws_min = p->ws_min; q.ws_max = p->ws_max; q.ws_inc = p->ws_inc; q.ws_dec = p->ws_dec;
        const int max_pc = p->max_per_component > 0 ? p->max_per_component : (N + K - 1) / K;
        if (p->presampled && n_old == 0) {
            // the previous call drew this iteration's samples behind its component update (below): only the bookkeeping is
            // left, and it rides in the target evaluation (riders.h) -- nothing before the component update reads what it
            // writes.  (Not in the dual sweep: at one 108 KB workgroup per CU that launch has no room for extra workgroups,
            // 16 riders made it 36 us instead of 24.)
            prep_later = q;
            prep_pending = true;
        } else {
            GMMVI_TRY(gmmvi_sample_components_prep(ctx, K, D, p->means, p->chols, p->offsets, N, max_pc, p->seed,
                                                   p->first_index, x, p->db_mapping, p->mapping_base, q));
        }
    }
    // ---- background + model density / gradient (sample_db.py:194-228, gmm.py:274-300) ------------------------------------
    // issued BEFORE the target evaluation, which does not depend on them: the merge of the model sweep's component-chunk
    // partials rides as extra workgroups in the target launch instead of a launch of its own (combine.h; same arithmetic)
    if (p->bg_packed == nullptr) {
        // nothing reused: the background components are the model's own -- one sweep for both mixtures
        ctx->defer_combine = true;
        ctx->prof_tag = "sweep_dual";
        int rc_dual = gmmvi_mixture_eval_dual(ctx, GMMVI_GAUSS, 0.f, K, D, p->packed, p->logw, p->bg_logw, x, N, a.ld, a.lq,
                                              a.qgrad, a.bg);
        ctx->defer_combine = false;
        ctx->prof_tag = nullptr;
        GMMVI_TRY(rc_dual);
    } else {
        // reused samples: the background mixture of the window = the mixture the reused samples came from (bg_K snapshot
        // components; its density is KNOWN for the reused samples -- bg_old, from the effective-sample-size step -- and is
        // evaluated for the new samples only) joined with the mixture of the new components over all active samples, each
        // with its share of the window (SampleDB.get_newest_samples takes the same route: optimization/sample_db.py)
        ctx->prof_tag = "sweep_background";
        int rc_bg = gmmvi_mixture_eval(ctx, GMMVI_GAUSS, 0.f, p->bg_K, D, p->bg_packed, p->bg_logw, x, N, nullptr, bg_a + n_old,
                                       nullptr);
        if (rc_bg == GMMVI_OK)
            rc_bg = gmmvi_mixture_eval(ctx, GMMVI_GAUSS, 0.f, K, D, p->packed, p->bg_logw_new, xa, Na, nullptr, bg_b, nullptr);
        ctx->prof_tag = nullptr;
        GMMVI_TRY(rc_bg);
        GMMVI_HIP_CHECK(ctx, hipMemcpyAsync(bg_a, p->bg_old, (size_t)n_old * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        GMMVI_TRY(gmmvi_logaddexp_f32(ctx, a.bg, bg_a, p->bg_log_share_old, bg_b, p->bg_log_share_new, (size_t)Na));
        ctx->defer_combine = true;
        ctx->prof_tag = "sweep_model";
        int rc_m = gmmvi_mixture_eval(ctx, GMMVI_GAUSS, 0.f, K, D, p->packed, p->logw, xa, Na, a.ld, a.lq, a.qgrad);
        ctx->defer_combine = false;
        ctx->prof_tag = nullptr;
        GMMVI_TRY(rc_m);
    }
    if (prep_pending) {
        ctx->riders.prep = prep_later;
        ctx->riders.prep_blocks = K < 64 ? K : 64;
    }
    if (p->target_kind == 1) {
        GMMVI_TRY(gmmvi_target_planar(ctx, D, p->planar_prior_std, p->planar_goals_count, p->planar_goals,
                                      p->planar_likelihood_std, x, N, p->db_tlp, p->db_tgrad));
    } else {
        ctx->prof_tag = "sweep_target";
        int rc_t = gmmvi_mixture_eval(ctx, p->target_family, p->target_nu, p->target_K, D, p->target_packed,
                                      p->target_logw, x, N, nullptr, p->db_tlp, p->db_tgrad);
        ctx->prof_tag = nullptr;
        GMMVI_TRY(rc_t);
    }
    GMMVI_TRY(gmmvi_flush_pending_combine(ctx));       // nothing left unless the target launch could not carry it
    GMMVI_TRY(gmmvi_flush_pending_riders(ctx));        // (the planar-robot target kernel carries no riders: their own launch)
    // ---- component update (gmmvi.py:165-169) -----------------------------------------------------------------------------
    // the Stein estimate stops at its partial slab; its last step (slab sum, Sigma^-1, normalisation) is the prologue of the
    // update kernel where that is instantiated, the stand-alone launch otherwise -- the same arithmetic either way
    SteinSlab slab{nullptr, nullptr, 0};
    GMMVI_TRY(gmmvi_stein_partials(ctx, K, D, p->packed, xa, Na, a.ld, a.qgrad, a.bg, tgrad_a, p->stein_flags, &slab));
    GMMVI_TRY(gmmvi_update_components_kl_from_slab(ctx, K, D, slab, Na, p->stein_flags, p->packed, a.H, a.g, p->means, p->chols,
                                                   p->stepsizes, p->temperature, p->l2_init, p->last_eta, p->l2, p->num_updates,
                                                   p->success_out ? p->success_out : a.success, p->packed_new));
    }
    if (phase == 1) return GMMVI_OK;
    // ---- weight update (gmmvi.py:172-173) ---------------------------------------------------------------------------------
    // (the merge of this sweep's log-density partials happens inside the expected-log-ratio kernel)
    ctx->defer_combine = true;
    ctx->prof_tag = "sweep_post";
    int rc_post = gmmvi_mixture_eval(ctx, GMMVI_GAUSS, 0.f, K, D, p->packed_new, p->logw, xa, Na, a.ld, a.lq, nullptr);
    ctx->defer_combine = false;
    ctx->prof_tag = nullptr;
    GMMVI_TRY(rc_post);
    if (p->presample_next && n_old == 0) {
        // the next iteration's draw from the UPDATED components rides in the expected-log-ratio launch, whose K workgroups leave most CUs idle (riders.h;
        // the draw needs neither the weights nor anything this launch writes).  Inside the post-update sweep it would queue behind
        // that launch's workgroups: every workgroup of a launch gets the launch's LDS size, 108 KB there
        const int max_pc = p->max_per_component > 0 ? p->max_per_component : (N + K - 1) / K;
        SampleJob& sj = ctx->riders.sample;
        sj.K = K; sj.D = D; sj.uniform_count = (long)K * max_pc == N ? max_pc : 0;
        sj.means = p->means; sj.chols = p->chols; sj.offsets = p->offsets;
        sj.seed = p->seed; sj.first_index = p->first_index + (uint64_t)N;
        sj.X = x + (size_t)N * D; sj.mapping = p->db_mapping + N; sj.mapping_base = p->mapping_base + K;
        ctx->riders.sample_blocks = K * ((max_pc + 255) / 256);
    }
    GMMVI_TRY(gmmvi_expected_log_ratios(ctx, K, Na, a.ld, a.bg, tlp_a, a.lq, p->temperature, p->logw,
                                        (p->stein_flags & GMMVI_SELF_NORMALIZED) ? 1 : 0, a.E, p->reward_next, nullptr));
    GMMVI_TRY(gmmvi_flush_pending_riders(ctx));        // (nothing is left normally)
    if (K > 1) {
        GMMVI_TRY(gmmvi_update_weights_internal(ctx, p->weight_update_mode == 0 ? 0 : 1, K, p->logw, a.E, p->wstate,
                                                p->temperature, nullptr, p->weight_slot));
    }
    return GMMVI_OK;
}


// ---- component shards: the iteration in four phases with an all-gather between them (include/gmmvi_hip.h) ---------------
namespace {
struct ShardArena { float *ld, *lq, *bg, *qgrad, *H, *g, *E; int32_t* success; };
}
extern "C" size_t gmmvi_sharded_scratch_floats(int K, int D, int N) {
    if (K < 1 || D < 1 || N < 1) return 0;
    return (size_t)K * N + 2 * (size_t)N + (size_t)N * D + (size_t)K * D * D + (size_t)K * D + 2 * (size_t)K + 64;
}
static void shard_arena(float* base, int K, int D, int N, ShardArena& a) {
    a.ld = base; base += (size_t)K * N;
    a.lq = base; base += N;
    a.bg = base; base += N;
    a.qgrad = base; base += (size_t)N * D;
    a.H = base; base += (size_t)K * D * D;
    a.g = base; base += (size_t)K * D;
    a.E = base; base += K;
    a.success = (int32_t*)base;
}

static int sharded_phase_body(gmmvi_ctx* ctx, const gmmvi_sharded_plan* p, int phase) {
    const int R = p->n_ranks, K = p->K, D = p->D, N = p->N;
    const int Nl = N / R, Kt = K * R;
    const size_t s1 = (size_t)Nl * (2 * D + 1) + 2 * (size_t)K, s2 = (size_t)N * (D + 2), s3 = (size_t)N;
    float* my1 = p->e1 + (size_t)p->rank * s1;         // x_loc | tlp_loc | tgrad_loc | E_loc | reward_loc
    float* x_loc = my1;
    float* tlp_loc = my1 + (size_t)Nl * D;
    float* tgrad_loc = tlp_loc + Nl;
    float* E_loc = tgrad_loc + (size_t)Nl * D;
    float* reward_loc = E_loc + K;
    float* my2 = p->e2 + (size_t)p->rank * s2;         // bg_part | lq_part | qgrad_part
    float* my3 = p->e3 + (size_t)p->rank * s3;
    float* logw_loc = p->logw_all + (size_t)p->rank * K;
    ShardArena a;
    shard_arena(p->scratch, K, D, N, a);
    const int max_pc = p->max_per_component > 0 ? p->max_per_component : (Nl + K - 1) / K;
    if (phase == 1) {
        // ---- the local draw and its target evaluation, written where the first exchange picks them up -----------------------
        if (!p->presampled)
            GMMVI_TRY(gmmvi_sample_components_bounded(ctx, K, D, p->means, p->chols, p->offsets, Nl, max_pc, p->seed, p->first_index, 0,
                                                      nullptr, x_loc, nullptr));
        if (p->target_kind == 1) {
            GMMVI_TRY(gmmvi_target_planar(ctx, D, p->planar_prior_std, p->planar_goals_count, p->planar_goals,
                                          p->planar_likelihood_std, x_loc, Nl, tlp_loc, tgrad_loc));
        } else {
            ctx->prof_tag = "sweep_target";
            int rc_t = gmmvi_mixture_eval(ctx, p->target_family, p->target_nu, p->target_K, D, p->target_packed, p->target_logw,
                                          x_loc, Nl, nullptr, tlp_loc, tgrad_loc);
            ctx->prof_tag = nullptr;
            GMMVI_TRY(rc_t);
        }
        return GMMVI_OK;
    }
    if (phase == 2) {
        // ---- de-interleave the first exchange (one rank: the parts ARE the gathered arrays when the caller aliased them) ------
        const bool aliased = R == 1 && p->x_all == x_loc && p->tlp_all == tlp_loc && p->tgrad_all == tgrad_loc &&
                             p->E_all == E_loc && p->reward_all == reward_loc;
        if (!aliased) {
            const size_t words[5] = {(size_t)Nl * D, (size_t)Nl, (size_t)Nl * D, (size_t)K, (size_t)K};
            void* const dst[5] = {p->x_all, p->tlp_all, p->tgrad_all, p->E_all, p->reward_all};
            GMMVI_TRY(gmmvi_unpack_gathered(ctx, p->e1, R, s1, 5, words, dst));
        }
        // ---- the previous iteration's weight step, replicated (weight_updater.py:56-100, gmm_wrapper.py:150-160) --------------
        if (p->has_pending) {
            GMMVI_HIP_CHECK(ctx, hipMemcpyAsync(p->reward_col_pending, p->reward_all, (size_t)Kt * sizeof(float),
                                                hipMemcpyDeviceToDevice, ctx->stream));
            if (Kt > 1)
                GMMVI_TRY(gmmvi_update_weights_internal(ctx, 0, Kt, p->logw_all, p->E_all, p->wstate, p->temperature, nullptr, nullptr));
        }
        // ---- stepsize rules: components local, weights over all components (iter_prep.h); they ride in the dual sweep when that
        // launch has room, else their own small launch
        {
            PrepArgs q{};
            q.K = K; q.cs_mode = p->component_stepsize_mode; q.stepsizes = p->stepsizes;
            q.reward_prev = p->reward_prev; q.reward_last = p->reward_last;
            q.cs_min = p->cs_min; q.cs_max = p->cs_max; q.cs_inc = p->cs_inc; q.cs_dec = p->cs_dec;
            q.ws_mode = p->weight_stepsize_mode; q.logw = p->logw_all; q.wstate = p->wstate;
            q.ws_min = p->ws_min; q.ws_max = p->ws_max; q.ws_inc = p->ws_inc; q.ws_dec = p->ws_dec;
            q.ws_K = Kt; q.ws_reward_last = p->reward_last_all;
            ctx->riders.prep = q;
            ctx->riders.prep_blocks = 1;
            GMMVI_TRY(gmmvi_flush_pending_riders(ctx));
        }
        // ---- dual sweep over the local components on all samples; its partials are this rank's part of the second exchange ----
        float* bg_out = R > 1 ? my2 : a.bg;
        float* lq_out = R > 1 ? my2 + N : a.lq;
        float* qg_out = R > 1 ? my2 + 2 * (size_t)N : a.qgrad;
        ctx->prof_tag = "sweep_dual";
        int rc_dual = gmmvi_mixture_eval_dual(ctx, GMMVI_GAUSS, 0.f, K, D, p->packed, logw_loc, p->bg_logw, p->x_all, N, a.ld,
                                              lq_out, qg_out, bg_out);
        ctx->prof_tag = nullptr;
        return rc_dual;
    }
    if (phase == 3) {
        if (R > 1) {
            // merge of the ranks' partials: log q and its gradient, the background as the second set of log values
            GMMVI_PROF(ctx, "mixture_combine");
            GMMVI_TRY(gmmvi_combine_partials_internal(ctx, R, N, D, p->e2 + N, p->e2 + 2 * (size_t)N, a.lq, a.qgrad, p->e2, a.bg,
                                                      (long)s2));
        }
        SteinSlab slab{nullptr, nullptr, 0};
        GMMVI_TRY(gmmvi_stein_partials(ctx, K, D, p->packed, p->x_all, N, a.ld, a.qgrad, a.bg, p->tgrad_all, p->stein_flags, &slab));
        GMMVI_TRY(gmmvi_update_components_kl_from_slab(ctx, K, D, slab, N, p->stein_flags, p->packed, a.H, a.g, p->means, p->chols,
                                                       p->stepsizes, p->temperature, p->l2_init, p->last_eta, p->l2, p->num_updates,
                                                       p->success_out ? p->success_out : a.success, p->packed_new));
        // post-update sweep: one rank leaves the merge of its chunk partials to the expected-log-ratio kernel (phase 4); with
        // more ranks the merged local log q is what travels
        ctx->defer_combine = R == 1;
        ctx->prof_tag = "sweep_post";
        int rc_post = gmmvi_mixture_eval(ctx, GMMVI_GAUSS, 0.f, K, D, p->packed_new, logw_loc, p->x_all, N, a.ld,
                                         R > 1 ? my3 : a.lq, nullptr);
        ctx->defer_combine = false;
        ctx->prof_tag = nullptr;
        return rc_post;
    }
    // ---- phase 4: expected log-ratios / rewards of the local components; with more ranks the gathered log q partials are merged
    // while they are read (the kernel's chunk-partial form: [R][N])
    if (R > 1) {
        GMMVI_TRY(gmmvi_flush_pending_combine(ctx));
        ctx->pending.R = R; ctx->pending.N = N; ctx->pending.D = D;
        ctx->pending.lp_parts = p->e3; ctx->pending.lp_out = a.lq;
    }
    if (p->presample_next) {
        // the next iteration's local draw from the updated components, into this rank's part of the first exchange buffer: nothing
        // reads that part any more (x was de-interleaved in phase 2; one rank with aliased views: the sweeps and the Stein estimate
        // are done, this launch reads log values only)
        SampleJob& sj = ctx->riders.sample;
        sj.K = K; sj.D = D; sj.uniform_count = (long)K * max_pc == Nl ? max_pc : 0;
        sj.means = p->means; sj.chols = p->chols; sj.offsets = p->offsets;
        sj.seed = p->seed; sj.first_index = p->first_index + (uint64_t)N;
        sj.X = x_loc; sj.mapping = nullptr; sj.mapping_base = 0;
        ctx->riders.sample_blocks = K * ((max_pc + 255) / 256);
    }
    GMMVI_TRY(gmmvi_expected_log_ratios(ctx, K, N, a.ld, a.bg, p->tlp_all, a.lq, p->temperature, logw_loc,
                                        (p->stein_flags & GMMVI_SELF_NORMALIZED) ? 1 : 0, E_loc, reward_loc, nullptr));
    GMMVI_TRY(gmmvi_flush_pending_riders(ctx));
    return GMMVI_OK;
}

extern "C" int gmmvi_train_iter_sharded_phase(gmmvi_ctx* ctx, const gmmvi_sharded_plan* p, int phase) {
    GMMVI_ARG_CHECK(ctx, ctx != nullptr && p != nullptr && phase >= 1 && phase <= 4);
    GMMVI_ARG_CHECK(ctx, p->n_ranks >= 1 && p->rank >= 0 && p->rank < p->n_ranks && p->K >= 1 && p->D >= 1 && p->D < GMMVI_MAX_DIM &&
                             p->N >= p->n_ranks && p->N % p->n_ranks == 0);
    GMMVI_ARG_CHECK(ctx, p->means && p->chols && p->packed && p->packed_new && p->stepsizes && p->last_eta && p->l2 && p->num_updates &&
                             p->logw_all && p->bg_logw && p->offsets && p->e1 && p->e2 && p->e3 && p->x_all && p->tlp_all &&
                             p->tgrad_all && p->E_all && p->reward_all && p->wstate && p->reward_prev && p->reward_last &&
                             p->reward_last_all && p->scratch);
    GMMVI_ARG_CHECK(ctx, !p->has_pending || p->reward_col_pending != nullptr);
    int rc = GMMVI_OK;
    if (phase != 4) rc = gmmvi_flush_pending_combine(ctx);     // (phase 4 consumes the merge phase 3 left for it)
    if (rc == GMMVI_OK) rc = sharded_phase_body(ctx, p, phase);
    if (rc != GMMVI_OK) {
        ctx->pending = CombineJob();
        ctx->riders.prep_blocks = ctx->riders.sample_blocks = 0;
        ctx->defer_combine = false;
        ctx->prof_tag = nullptr;
    }
    return rc;
}
