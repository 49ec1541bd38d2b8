// libefg_hip.so: error reporting, version, capture-stream helpers (host only).
#include "common.h"

namespace efg {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace efg

extern "C" const char* efg_last_error(void) { return efg::g_err; }
extern "C" const char* efg_version(void) { return "efg_hip 0.1 gfx950"; }

// A stream of the caller's own for graph captures.  A capture that gets invalidated leaves its stream in
// hipStreamCaptureStatusInvalidated for good on ROCm 7 (hipStreamEndCapture reports the error and does not end it;
// scripts/repro/invalidated_stream.py), so captures must not run on a stream that anything else will be handed later
// (PyTorch's pool streams are): the caller creates one here and destroys it if its capture went wrong.
extern "C" int efg_capture_stream_create(void** stream_out) {
  EFG_CHECK_ARG(stream_out != nullptr, "efg_capture_stream_create: stream_out is NULL");
  hipStream_t s = nullptr;
  EFG_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream_out = s;
  return EFG_OK;
}

extern "C" int efg_capture_stream_destroy(void* stream) {
  EFG_CHECK_ARG(stream != nullptr, "efg_capture_stream_destroy: the NULL stream is not the caller's to destroy");
  EFG_HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return EFG_OK;
}
