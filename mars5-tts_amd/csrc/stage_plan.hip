// Stage-level entry points (include/mars5_hip.h: m5_nar_step, m5_ar_decode_step, m5_stage_run): a whole stage of the hot path --
// the ~190 launches of one NAR reverse step (reference mars5/diffuser.py:345-394 around model.py:264-343), the launches of one
// AR decode step (nn_future.py:369-398 + ar_generate.py:74-121) -- enqueued by ONE call from a caller-filled plan.
//
// A plan is data: an array of M5PlanOp {entry point, arguments} in launch order plus an arena holding copies of the argument
// structures (M5AttnArgs, M5DeferredLN, ...).  The host engine fills it once per session (every argument of a step is a device
// pointer, a stride or a size -- the step index, positions and RNG state live in device memory), after which a step is one C
// call; the composition no longer lives in the host language, which only allocates and (optionally) captures the call in a
// hipGraph.  Argument slots: integers by value, pointers by address, floats as their IEEE bits, structure pointers as
// (arena byte offset + 1), 0 = NULL.  The executor unpacks a slot list against the REAL prototype of the entry point (template
// over the function type: a mismatch between plan and prototype is a compile error here or an argument-count error at run time).
#include "common.h"
#include <tuple>
#include <type_traits>
#include <utility>

namespace {

template <typename T>
struct SlotArg {
    static bool get(int64_t slot, const M5StagePlan* plan, T& out) {
        if constexpr (std::is_pointer_v<T>) {
            using P = std::remove_cv_t<std::remove_pointer_t<T>>;
            if constexpr (std::is_class_v<P>) {                       // an argument structure: lives in the arena
                if (slot == 0) { out = nullptr; return true; }
                const int64_t off = slot - 1;
                if (off < 0 || (off & 7) || off + (int64_t)sizeof(P) > plan->arena_bytes || !plan->arena) return false;
                out = reinterpret_cast<T>(const_cast<unsigned char*>(plan->arena) + off);
                return true;
            } else {
                out = reinterpret_cast<T>((uintptr_t)slot);            // device (or opaque) pointer
                return true;
            }
        } else if constexpr (std::is_floating_point_v<T>) {
            const uint32_t bits = (uint32_t)slot;
            float f;
            __builtin_memcpy(&f, &bits, 4);
            out = (T)f;
            return true;
        } else {
            out = (T)slot;
            return true;
        }
    }
};

// call fn(slots..., stream): the LAST parameter of every launch entry point is the stream
template <size_t I, size_t N, typename T>
bool fill_arg(T& out, const M5PlanOp& op, const M5StagePlan* plan, void* stream) {
    if constexpr (I + 1 == N) {
        static_assert(std::is_same_v<T, void*>, "the last parameter of a launch entry point is the stream");
        out = stream;
        return true;
    } else {
        return SlotArg<T>::get(op.a[I], plan, out);
    }
}
template <typename R, typename... A, size_t... I>
int invoke_impl(R (*fn)(A...), const M5PlanOp& op, const M5StagePlan* plan, void* stream, std::index_sequence<I...>) {
    constexpr size_t N = sizeof...(A);
    if (op.n_args != (int)N - 1) return M5_ERR_ARG;
    std::tuple<std::remove_cv_t<A>...> args{};
    const bool ok = (fill_arg<I, N>(std::get<I>(args), op, plan, stream) && ...);
    if (!ok) return M5_ERR_ARG;
    return std::apply(fn, args);
}
template <typename R, typename... A>
int invoke(R (*fn)(A...), const M5PlanOp& op, const M5StagePlan* plan, void* stream) {
    static_assert(sizeof...(A) - 1 <= M5_PLAN_MAX_ARGS, "M5_PLAN_MAX_ARGS");
    return invoke_impl(fn, op, plan, stream, std::index_sequence_for<A...>{});
}

constexpr unsigned KIND_NAR = 1u, KIND_AR = 2u;

int run_op(const M5PlanOp& op, const M5StagePlan* plan, void* stream, unsigned allowed) {
#define M5_OP(FN, NAME, KINDS) case FN: if (!((KINDS) & allowed)) return M5_ERR_ARG; return invoke(&NAME, op, plan, stream);
    switch (op.fn) {
        M5_OP(M5_FN_GEMM, m5_gemm, KIND_NAR | KIND_AR)
        M5_OP(M5_FN_GEMM_EX, m5_gemm_ex, KIND_NAR)
        M5_OP(M5_FN_LAYERNORM, m5_layernorm, KIND_NAR)
        M5_OP(M5_FN_LAYERNORM_TWICE, m5_layernorm_twice, KIND_NAR)
        M5_OP(M5_FN_LAYERNORM_MEAN, m5_layernorm_mean, KIND_NAR)
        M5_OP(M5_FN_RMSNORM, m5_rmsnorm, KIND_AR)
        M5_OP(M5_FN_ATTENTION, m5_attention, KIND_NAR | KIND_AR)
        M5_OP(M5_FN_GATHER_ROWS, m5_gather_rows, KIND_NAR | KIND_AR)
        M5_OP(M5_FN_CHUNKED_EMBED, m5_chunked_embed, KIND_NAR)
        M5_OP(M5_FN_XATTN_ABSORB, m5_xattn_absorb, KIND_NAR)
        M5_OP(M5_FN_XATTN_SCORES, m5_xattn_scores, KIND_NAR)
        M5_OP(M5_FN_XATTN_SCORES_EX, m5_xattn_scores_ex, KIND_NAR)
        M5_OP(M5_FN_NAR_UNIFORMS, m5_nar_uniforms, KIND_NAR)
        M5_OP(M5_FN_NAR_SAMPLE, m5_nar_sample, KIND_NAR)
        M5_OP(M5_FN_ADD_INT, m5_add_int, KIND_NAR | KIND_AR)
        M5_OP(M5_FN_COPY_D2D, m5_copy_d2d, KIND_NAR | KIND_AR)
        M5_OP(M5_FN_AR_GEMV, m5_ar_gemv, KIND_AR)
        M5_OP(M5_FN_AR_ATTN_DECODE, m5_ar_attn_decode, KIND_AR)
        M5_OP(M5_FN_AR_LAYERS_PERSISTENT, m5_ar_layers_persistent, KIND_AR)
        M5_OP(M5_FN_AR_SAMPLE, m5_ar_sample, KIND_AR)
        M5_OP(M5_FN_AR_ROPE_CACHE_BATCH, m5_ar_rope_cache_batch, KIND_AR)
        M5_OP(M5_FN_AR_QKV_ROPE_BATCH, m5_ar_qkv_rope_batch, KIND_AR)
        M5_OP(M5_FN_AR_ATTN_COMBINE_BATCH, m5_ar_attn_combine_batch, KIND_AR)
        default: return M5_ERR_ARG;
    }
#undef M5_OP
}

int run_plan(const M5StagePlan* plan, void* stream, unsigned allowed) {
    if (!plan || plan->n_ops < 0 || (plan->n_ops > 0 && !plan->ops) || plan->arena_bytes < 0 || (plan->arena_bytes > 0 && !plan->arena)) return M5_ERR_ARG;
    if (plan->failed_op) *plan->failed_op = -1;
    for (int i = 0; i < plan->n_ops; ++i) {
        const int rc = run_op(plan->ops[i], plan, stream, allowed);
        if (rc != M5_OK) {                                          // the launches before op i are enqueued; the caller decides (a
            if (plan->failed_op) *plan->failed_op = i;              // captured graph is discarded, an eager step is an error)
            return rc;
        }
    }
    return M5_OK;
}

}  // namespace

extern "C" int m5_copy_d2d(void* dst, const void* src, int64_t bytes, void* stream) {
    if (!dst || !src || bytes < 0) return M5_ERR_ARG;
    if (bytes == 0) return M5_OK;
    return hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}

extern "C" int m5_stage_run(const M5StagePlan* plan, void* stream) { return run_plan(plan, stream, KIND_NAR | KIND_AR); }
extern "C" int m5_nar_step(const M5StagePlan* plan, void* stream) { return run_plan(plan, stream, KIND_NAR); }
extern "C" int m5_ar_decode_step(const M5StagePlan* plan, void* stream) { return run_plan(plan, stream, KIND_AR); }
