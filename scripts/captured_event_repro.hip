// Root-cause probe for the round-2 RCCL-watchdog abort (hipErrorCapturedEvent).
//
// Question: does hipEventQuery() fail for an event that was recorded EAGERLY (outside any capture) on a
// stream that has SINCE entered a capture?  c10d's ProcessGroupNCCL records every Work's end event on ITS
// internal stream; works of the eager warm-up steps stay in the watchdog's list until its next 100-ms pass.
// If the same internal stream is then forked into a HIP-graph capture (async_op=True inside torch.cuda.graph),
// a "yes" here means the watchdog's poll of a long-finished warm-up Work aborts the process.
//
//   hipcc --offload-arch=gfx950 -O2 -o captured_event_repro scripts/captured_event_repro.hip -lpthread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FATAL %s -> %s\n", #x, hipGetErrorName(e_)); return 2; } } while (0)

__global__ void touch(int* p) { if (p) atomicAdd(p, 1); }

static const char* q(hipEvent_t e) {
    hipError_t r = hipEventQuery(e);
    (void)hipGetLastError();
    return hipGetErrorName(r);
}

static void q_thread(hipEvent_t e, const char** out) {
    hipSetDevice(0);
    *out = q(e);
}

int main() {
    int* d;
    CK(hipMalloc(&d, 4));
    CK(hipMemset(d, 0, 4));
    hipStream_t cap, side, other;
    CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
    hipEvent_t e_side, e_other, fork, join;
    CK(hipEventCreateWithFlags(&e_side, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e_other, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));

    // "warm-up": eager work + end events on `side` (c10d's internal stream) and on `other` (never captured)
    touch<<<1, 64, 0, side>>>(d);
    CK(hipEventRecord(e_side, side));
    touch<<<1, 64, 0, other>>>(d);
    CK(hipEventRecord(e_other, other));
    CK(hipDeviceSynchronize());
    printf("before capture:            e_side=%s e_other=%s\n", q(e_side), q(e_other));

    // capture on `cap`, thread-local mode (what GraphedTrainStep uses)
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    touch<<<1, 64, 0, cap>>>(d);
    const char *a = nullptr, *b = nullptr;
    { std::thread t(q_thread, e_side, &a); t.join(); }
    { std::thread t(q_thread, e_other, &b); t.join(); }
    printf("capturing, side NOT forked: e_side=%s e_other=%s   (watchdog thread)\n", a, b);

    // fork `side` into the capture (what an async_op=True collective does inside the capture)
    CK(hipEventRecord(fork, cap));
    CK(hipStreamWaitEvent(side, fork, 0));
    touch<<<1, 64, 0, side>>>(d);
    { std::thread t(q_thread, e_side, &a); t.join(); }
    { std::thread t(q_thread, e_other, &b); t.join(); }
    printf("capturing, side FORKED:     e_side=%s e_other=%s   (watchdog thread; e_side was recorded EAGERLY)\n", a, b);
    hipStreamCaptureStatus st;
    hipError_t sr = hipStreamIsCapturing(cap, &st);
    printf("capture status after the poll: %s status=%d (1 = active, 2 = invalidated)\n", hipGetErrorName(sr), (int)st);

    CK(hipEventRecord(join, side));
    CK(hipStreamWaitEvent(cap, join, 0));
    hipGraph_t g = nullptr;
    hipError_t er = hipStreamEndCapture(cap, &g);
    (void)hipGetLastError();
    printf("end capture: %s graph=%p\n", hipGetErrorName(er), (void*)g);
    printf("after capture:             e_side=%s e_other=%s\n", q(e_side), q(e_other));
    return 0;
}
