// Host launch path in C: falnet_replay issues a pre-built sequence of C-ABI launches, event records and stream waits from ONE call.
//
// The step's launch order, streams and dependencies are static per plan (fal_net_amd/models/FAL_netB.py): ~330 launches and ~50
// event operations per Stage-1 step, each a ctypes call from Python (~6 us of host time apiece, 2-4.5 ms per step).  A plan records
// the sequence once -- the SAME entry points with the SAME arguments, the same streams, the same events -- and replays it here: eager
// semantics are kept (every launch is an ordinary launch on its stream; nothing is captured into a graph), only the issuing loop moves
// from the interpreter into this file.  Reference counterpart: none (the reference's launches are issued by aten, Train_Stage1_K.py:233-262).
//
// A command names its entry point by index (falnet_replay_op_index), carries the integer / pointer arguments in declaration order in
// iarg[], the floating-point ones in farg[], and the index of its stream in the caller's stream table; the trailing `void* stream`
// parameter every launch function has is filled in here.  The typed thunks below are generated from the prototypes of
// include/falnet_hip.h, so a signature change is a compile error, not a silent mis-call.
#include <string.h>
#include <tuple>
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

template <typename T>
inline T take(const falnet_cmd_t& c, int& ii, int& fi) {
    if constexpr (std::is_floating_point<T>::value) return (T)c.farg[fi++];
    else if constexpr (std::is_pointer<T>::value) return (T)(uintptr_t)c.iarg[ii++];
    else return (T)(int64_t)c.iarg[ii++];
}

template <typename F>
struct FnTraits;
template <typename... Args>
struct FnTraits<int (*)(Args...)> {
    using tuple = std::tuple<Args...>;
    static constexpr size_t N = sizeof...(Args);
};

template <auto FN>
struct OpThunk {  // FN: int (*)(A..., void* stream)
    using Tup = typename FnTraits<decltype(FN)>::tuple;
    static constexpr size_t N = FnTraits<decltype(FN)>::N;
    static_assert(N >= 1 && std::is_same<std::tuple_element_t<N - 1, Tup>, void*>::value, "launch functions end in `void* stream`");
    template <size_t... I>
    static int call_impl(const falnet_cmd_t& c, void* stream, std::index_sequence<I...>) {
        int ii = 0, fi = 0;
        std::tuple<std::tuple_element_t<I, Tup>...> args{take<std::tuple_element_t<I, Tup>>(c, ii, fi)...};  // braced: evaluated left to right
        (void)ii, (void)fi;
        return FN(std::get<I>(args)..., stream);
    }
    static int call(const falnet_cmd_t& c, void* stream) { return call_impl(c, stream, std::make_index_sequence<N - 1>{}); }
    template <size_t... I>
    static constexpr int count_flt(std::index_sequence<I...>) {
        return (0 + ... + (std::is_floating_point<std::tuple_element_t<I, Tup>>::value ? 1 : 0));
    }
    static constexpr int nflt() { return count_flt(std::make_index_sequence<N - 1>{}); }
    static constexpr int nint() { return (int)(N - 1) - nflt(); }
    // a command has FALNET_CMD_MAX_INT integer / pointer slots and FALNET_CMD_MAX_FLT floating-point ones: an entry point that needs more is a
    // compile error here, not an out-of-bounds read of the next command at run time (falnet_augment_normalize takes eight floats)
    static_assert(nint() <= FALNET_CMD_MAX_INT && nflt() <= FALNET_CMD_MAX_FLT, "entry point does not fit a falnet_cmd_t: widen iarg / farg");
};
static_assert(sizeof(((falnet_cmd_t*)0)->iarg) / sizeof(uint64_t) == FALNET_CMD_MAX_INT && sizeof(((falnet_cmd_t*)0)->farg) / sizeof(double) == FALNET_CMD_MAX_FLT,
              "falnet_cmd_t slot counts");

struct Op {
    const char* name;
    int (*call)(const falnet_cmd_t&, void*);
    int nint, nflt;
};
#define FALNET_OP(fn) {#fn, &OpThunk<&fn>::call, OpThunk<&fn>::nint(), OpThunk<&fn>::nflt()}
const Op g_ops[] = {
    FALNET_OP(falnet_conv2d), FALNET_OP(falnet_conv3x3_c3), FALNET_OP(falnet_conv2d_multi), FALNET_OP(falnet_wgrad), FALNET_OP(falnet_wgrad_reduce),
    FALNET_OP(falnet_bias_grad), FALNET_OP(falnet_pack_weights_batched), FALNET_OP(falnet_adam_pack_batched), FALNET_OP(falnet_adam_ranges),
    FALNET_OP(falnet_adam_tick), FALNET_OP(falnet_pack_up2_batched), FALNET_OP(falnet_wgrad_reduce_batched), FALNET_OP(falnet_bias_grad_batched),
    FALNET_OP(falnet_bias_grad_batched_det), FALNET_OP(falnet_pack_weights), FALNET_OP(falnet_nchw_to_nhwc), FALNET_OP(falnet_nhwc_to_nchw),
    FALNET_OP(falnet_upsample_bwd), FALNET_OP(falnet_wgrad_const_plane), FALNET_OP(falnet_maxpool2_fwd), FALNET_OP(falnet_maxpool2_bwd),
    FALNET_OP(falnet_act_bwd), FALNET_OP(falnet_med_head_fwd), FALNET_OP(falnet_med_head_bwd), FALNET_OP(falnet_med_head_bwd_nhwc),
    FALNET_OP(falnet_med_masks_fwd), FALNET_OP(falnet_med_maskr_acfalse_fwd), FALNET_OP(falnet_l1_fwd), FALNET_OP(falnet_l1_bwd),
    FALNET_OP(falnet_mse_fwd), FALNET_OP(falnet_mse_bwd), FALNET_OP(falnet_smooth_fwd), FALNET_OP(falnet_smooth_bwd), FALNET_OP(falnet_l1_fwd_bwd),
    FALNET_OP(falnet_l1_fwd_bwd_add), FALNET_OP(falnet_mse_fwd_bwd), FALNET_OP(falnet_smooth_fwd_bwd), FALNET_OP(falnet_step_scalars),
    FALNET_OP(falnet_mask_mix), FALNET_OP(falnet_adam_step), FALNET_OP(falnet_adam_step_dev), FALNET_OP(falnet_grad_guard),
    FALNET_OP(falnet_adam_step_guarded), FALNET_OP(falnet_loss_scale_update), FALNET_OP(falnet_loss_seeds), FALNET_OP(falnet_occlusion_mask),
    FALNET_OP(falnet_mirror_weight), FALNET_OP(falnet_hflip), FALNET_OP(falnet_rowmax), FALNET_OP(falnet_gemm_f32_small),
    FALNET_OP(falnet_resize_planar), FALNET_OP(falnet_disp_prologue), FALNET_OP(falnet_resample_u8), FALNET_OP(falnet_augment_normalize),
    FALNET_OP(falnet_fill_f32), FALNET_OP(falnet_copy_bytes), FALNET_OP(falnet_spin),
};
constexpr int g_nops = (int)(sizeof(g_ops) / sizeof(g_ops[0]));

}  // namespace

extern "C" int falnet_replay_op_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < g_nops; ++i)
        if (strcmp(g_ops[i].name, name) == 0) return i;
    return -1;
}

extern "C" int falnet_replay_op_args(int op, int* nint, int* nflt) {
    FALNET_CHECK_ARG(op >= 0 && op < g_nops && nint && nflt, "replay_op_args: unknown op %d", op);
    *nint = g_ops[op].nint;
    *nflt = g_ops[op].nflt;
    return 0;
}

extern "C" int falnet_replay(const falnet_cmd_t* cmds, int n, void* const* streams, int nstreams, void* const* events, int nevents, int* failed_at) {
    FALNET_CHECK_ARG(cmds && n >= 0 && streams && nstreams > 0, "replay: bad argument");
    // Every stream of a replay must belong to ONE device: checked once per call (hipStreamGetDevice is a table lookup), then that device is
    // selected once instead of per launch (falnet_enter_stream is skipped while depth > 0).  With a NULL main stream (the thread's current
    // device decides where it runs) the per-launch selection stays on.
    hipDevice_t dev0 = -1;
    for (int s = 0; s < nstreams; ++s) {
        if (!streams[s]) continue;
        hipDevice_t d;
        if (hipStreamGetDevice((hipStream_t)streams[s], &d) != hipSuccess) {
            (void)hipGetLastError();
            falnet_set_error("replay: stream %d is not a valid stream handle", s);
            return -1;
        }
        if (dev0 >= 0 && d != dev0) {
            falnet_set_error("replay: streams of devices %d and %d in one sequence", (int)dev0, (int)d);
            return -1;
        }
        dev0 = d;
    }
    const bool pin = streams[0] != nullptr;
    if (pin) falnet_enter_stream(streams[0]);
    struct Depth {
        int* d;
        bool on;
        explicit Depth(bool o) : d(falnet_replay_depth()), on(o) { if (on) ++*d; }
        ~Depth() { if (on) --*d; }
    } depth_guard(pin);
    for (int i = 0; i < n; ++i) {
        const falnet_cmd_t& c = cmds[i];
        int rc = 0;
        if (c.stream < 0 || c.stream >= nstreams) {
            falnet_set_error("replay: command %d names stream %d of %d", i, c.stream, nstreams);
            rc = -1;
        } else if (c.op == FALNET_CMD_RECORD || c.op == FALNET_CMD_WAIT) {
            if (!events || c.event < 0 || c.event >= nevents || !events[c.event]) {
                falnet_set_error("replay: command %d names event %d of %d", i, c.event, nevents);
                rc = -1;
            } else {
                const hipError_t e = c.op == FALNET_CMD_RECORD ? hipEventRecord((hipEvent_t)events[c.event], (hipStream_t)streams[c.stream])
                                                               : hipStreamWaitEvent((hipStream_t)streams[c.stream], (hipEvent_t)events[c.event], 0);
                if (e != hipSuccess) {
                    falnet_set_error("replay: command %d (%s): %s", i, c.op == FALNET_CMD_RECORD ? "event record" : "stream wait", hipGetErrorString(e));
                    rc = (int)e;
                }
            }
        } else if (c.op >= 0 && c.op < g_nops) {
            if (c.nint < 0 || c.nint > FALNET_CMD_MAX_INT || c.nflt < 0 || c.nflt > FALNET_CMD_MAX_FLT || c.nint != g_ops[c.op].nint || c.nflt != g_ops[c.op].nflt) {
                falnet_set_error("replay: command %d (%s) carries %d + %d arguments, the entry point takes %d + %d", i, g_ops[c.op].name, c.nint, c.nflt,
                                 g_ops[c.op].nint, g_ops[c.op].nflt);
                rc = -1;
            } else {
                rc = g_ops[c.op].call(c, streams[c.stream]);
            }
        } else {
            falnet_set_error("replay: command %d has unknown op %d", i, c.op);
            rc = -1;
        }
        if (rc != 0) {
            if (failed_at) *failed_at = i;
            return rc;
        }
    }
    if (failed_at) *failed_at = -1;
    return 0;
}

// the two aten launches a static step otherwise needs (gradient-buffer fill, input copy), as entry points so that they can be replayed
extern "C" int falnet_fill_f32(float* p, int64_t n, float value, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(p && n >= 0, "fill_f32: bad argument");
    if (n == 0) return 0;
    const hipError_t e = value == 0.f ? hipMemsetAsync(p, 0, (size_t)n * 4, (hipStream_t)stream)
                                      : hipMemsetD32Async((hipDeviceptr_t)p, __builtin_bit_cast(int, value), (size_t)n, (hipStream_t)stream);
    if (e != hipSuccess) {
        falnet_set_error("fill_f32: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int falnet_copy_bytes(void* dst, const void* src, int64_t nbytes, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(dst && src && nbytes >= 0, "copy_bytes: bad argument");
    if (nbytes == 0) return 0;
    const hipError_t e = hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) {
        falnet_set_error("copy_bytes: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
