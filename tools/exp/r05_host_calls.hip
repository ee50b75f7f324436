// Round 5: what the runtime's calls cost the calling thread on this box -- a kernel launch, an event record, a stream wait, a
// four-kernel chain as stream launches and as one hipGraphLaunch.  Prints one JSON line.
//   hipcc --offload-arch=gfx950 -O2 -o tools/exp/_build/r05_host_calls tools/exp/r05_host_calls.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void small_kernel(const uint4* a, uint4* b, const uint4* t, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
    (void)t;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    uint4 *a, *b;
    const unsigned n = 1u << 16;
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16));
    hipEvent_t ev[4];
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int reps = 2000;
    auto chain = [&](hipStream_t s) {
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(small_kernel, dim3(n / 256), dim3(256), 0, s, a, b, a, n);
    };
    for (int k = 0; k < 200; ++k) chain(s0);
    CK(hipStreamSynchronize(s0));
    // (a) launches into a busy queue
    double t0 = now_us();
    for (int k = 0; k < reps; ++k) chain(s0);
    const double launch_us = (now_us() - t0) / (reps * 4);
    CK(hipStreamSynchronize(s0));
    // (b) launch + event record
    t0 = now_us();
    for (int k = 0; k < reps; ++k) { chain(s0); CK(hipEventRecord(ev[0], s0)); }
    const double with_record = (now_us() - t0) / reps;
    CK(hipStreamSynchronize(s0));
    // (c) two streams, chain alternating with record + wait
    t0 = now_us();
    for (int k = 0; k < reps; ++k) {
        hipStream_t s = (k & 1) ? s1 : s0;
        CK(hipStreamWaitEvent(s, ev[(k & 1) ^ 1], 0));
        chain(s);
        CK(hipEventRecord(ev[k & 1], s));
    }
    const double two_streams = (now_us() - t0) / reps;
    CK(hipStreamSynchronize(s0));
    CK(hipStreamSynchronize(s1));
    // (d) the chain as a graph
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    chain(s0);
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int k = 0; k < 200; ++k) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    t0 = now_us();
    for (int k = 0; k < reps; ++k) CK(hipGraphLaunch(ge, s0));
    const double graph_us = (now_us() - t0) / reps;
    CK(hipStreamSynchronize(s0));
    // GPU time of the chain both ways
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms_stream = 0, ms_graph = 0;
    CK(hipEventRecord(e0, s0));
    for (int k = 0; k < reps; ++k) chain(s0);
    CK(hipEventRecord(e1, s0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_stream, e0, e1));
    CK(hipEventRecord(e0, s0));
    for (int k = 0; k < reps; ++k) CK(hipGraphLaunch(ge, s0));
    CK(hipEventRecord(e1, s0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_graph, e0, e1));
    printf("{\"launch_us\": %.2f, \"chain4_plus_record_us\": %.2f, \"chain4_two_streams_wait_record_us\": %.2f, \"graph_launch_chain4_us\": %.2f, "
           "\"gpu_chain4_stream_us\": %.2f, \"gpu_chain4_graph_us\": %.2f}\n",
           launch_us, with_record, two_streams, graph_us, ms_stream * 1e3 / reps, ms_graph * 1e3 / reps);
    return 0;
}
