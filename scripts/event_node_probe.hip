// Probe: event-record nodes inside a captured HIP graph, added explicitly (works on the HIP 7.0 runtime PyTorch bundles and on
// ROCm 7.2; hipEventRecordWithFlags(hipEventRecordExternal) is rejected by 7.0).  hipcc --offload-arch=gfx950 -O2 it, run on the box;
// LD_PRELOAD=<torch>/lib/libamdhip64.so selects the bundled runtime.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f; p[threadIdx.x] = a; }
#define CK(x) do { hipError_t e = (x); printf("%-70s -> %s\n", #x, hipGetErrorString(e)); } while (0)
static void rec_node(hipStream_t st, hipEvent_t ev) {
    hipStreamCaptureStatus cs; unsigned long long id; hipGraph_t g; const hipGraphNode_t* deps; size_t nd;
    CK(hipStreamGetCaptureInfo_v2(st, &cs, &id, &g, &deps, &nd));
    hipGraphNode_t node;
    CK(hipGraphAddEventRecordNode(&node, g, deps, nd, ev));
    CK(hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies));
}
int main(int argc, char** argv) {
    float* d; CK(hipMalloc(&d, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    spin<<<1, 64, 0, st>>>(d, 1000);
    rec_node(st, e0);
    spin<<<1, 64, 0, st>>>(d, 100000);
    CK(hipGetLastError());
    rec_node(st, e1);
    spin<<<1, 64, 0, st>>>(d, 1000);
    hipGraph_t g; CK(hipStreamEndCapture(st, &g));
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) {
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        float ms = -1; CK(hipEventElapsedTime(&ms, e0, e1)); printf("replay %d: %.3f ms\n", r, ms);
    }
    return 0;
}
