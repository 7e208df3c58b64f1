// ABI version, hipGraph capture helpers and HIP-event timing.
#include "common.h"

extern "C" int m5_version(void) { return 1; }
extern "C" const char* m5_build_info(void) { return "mars5_hip gfx950 wave64 (f32/f16/bf16 MFMA) built " __DATE__; }

extern "C" int m5_graph_begin(void* stream) {
    if (!stream) return M5_ERR_ARG;   // the legacy default stream cannot be captured
    return hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}

extern "C" int m5_graph_end(void* stream, void** graph_exec) {
    if (!stream || !graph_exec) return M5_ERR_ARG;
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || !g) return M5_ERR_LAUNCH;
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return M5_ERR_LAUNCH;
    *graph_exec = (void*)ex;
    return M5_OK;
}

extern "C" int m5_graph_launch(void* graph_exec, void* stream) {
    if (!graph_exec) return M5_ERR_ARG;
    return hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}

extern "C" int m5_graph_destroy(void* graph_exec) {
    if (!graph_exec) return M5_ERR_ARG;
    return hipGraphExecDestroy((hipGraphExec_t)graph_exec) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}

extern "C" int m5_event_create(void** ev) {
    if (!ev) return M5_ERR_ARG;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return M5_ERR_LAUNCH;
    *ev = (void*)e;
    return M5_OK;
}
extern "C" int m5_event_record(void* ev, void* stream) {
    if (!ev) return M5_ERR_ARG;
    return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}
extern "C" int m5_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    if (!ev_start || !ev_stop || !ms) return M5_ERR_ARG;
    if (hipEventSynchronize((hipEvent_t)ev_stop) != hipSuccess) return M5_ERR_LAUNCH;
    return hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}
extern "C" int m5_event_destroy(void* ev) {
    if (!ev) return M5_ERR_ARG;
    return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}
