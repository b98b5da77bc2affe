// runtime.hip — error reporting, device query, hipGraph capture and hipEvent timing helpers of libmgld_hip.
// The reference has no native runtime (single Python thread + torch stream); this is the thin
// MI355X-side replacement: the host captures one reverse-diffusion step into a hipGraph and replays it.
#include "common.h"
#include <stdio.h>
#include <string.h>

static char g_err[512] = "";

void mgld_set_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

int mgld_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mgld_set_error(what, e);
    return MGLD_E_LAUNCH;
  }
  return MGLD_OK;
}

#define HIP_TRY(expr, what)                 \
  do {                                      \
    hipError_t _e = (expr);                 \
    if (_e != hipSuccess) {                 \
      mgld_set_error(what, _e);             \
      return MGLD_E_LAUNCH;                 \
    }                                       \
  } while (0)

extern "C" int mgld_version(void) { return 100; }
extern "C" const char* mgld_last_error(void) { return g_err; }

extern "C" int mgld_device_info(int device, int64_t* out4) {
  MGLD_REQUIRE(out4, "device_info: null");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
  out4[0] = prop.multiProcessorCount;
  out4[1] = (int64_t)prop.maxSharedMemoryPerMultiProcessor;
  out4[2] = prop.clockRate;
  int arch = 0;
  const char* g = strstr(prop.gcnArchName, "gfx");
  if (g) arch = atoi(g + 3);
  out4[3] = arch;
  return MGLD_OK;
}

extern "C" int mgld_graph_begin(void* stream) {
  HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed), "hipStreamBeginCapture");
  return MGLD_OK;
}

extern "C" int mgld_graph_end(void* stream, void** graph_exec_out) {
  MGLD_REQUIRE(graph_exec_out, "graph_end: null");
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &graph), "hipStreamEndCapture");
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    mgld_set_error("hipGraphInstantiate", e);
    return MGLD_E_LAUNCH;
  }
  *graph_exec_out = (void*)exec;
  return MGLD_OK;
}

extern "C" int mgld_graph_launch(void* graph_exec, void* stream) {
  HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream), "hipGraphLaunch");
  return MGLD_OK;
}

extern "C" int mgld_graph_destroy(void* graph_exec) {
  if (graph_exec) HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec), "hipGraphExecDestroy");
  return MGLD_OK;
}

extern "C" int mgld_event_create(void** ev_out) {
  MGLD_REQUIRE(ev_out, "event_create: null");
  hipEvent_t ev;
  HIP_TRY(hipEventCreate(&ev), "hipEventCreate");
  *ev_out = (void*)ev;
  return MGLD_OK;
}
extern "C" int mgld_event_record(void* ev, void* stream) {
  HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream), "hipEventRecord");
  return MGLD_OK;
}
extern "C" int mgld_event_sync(void* ev) {
  HIP_TRY(hipEventSynchronize((hipEvent_t)ev), "hipEventSynchronize");
  return MGLD_OK;
}
extern "C" int mgld_event_elapsed_ms(void* a, void* b, float* ms_out) {
  MGLD_REQUIRE(ms_out, "event_elapsed: null");
  HIP_TRY(hipEventElapsedTime(ms_out, (hipEvent_t)a, (hipEvent_t)b), "hipEventElapsedTime");
  return MGLD_OK;
}
extern "C" int mgld_event_destroy(void* ev) {
  if (ev) HIP_TRY(hipEventDestroy((hipEvent_t)ev), "hipEventDestroy");
  return MGLD_OK;
}
