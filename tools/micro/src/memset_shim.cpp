// LD_PRELOAD shim: log every hipMemsetAsync / hipMemsetD8Async / hipMemsetD32Async / hipMemset2DAsync issued while its stream is CAPTURING
// (memset nodes of a HIP graph).  On this stack (ROCm 7.2.0, torch 2.10.0+rocm7.0) a captured memset node works in the first replay of
// the graph and writes garbage from the second on (tools/micro/memset_graph_check.py), so a captured step must not contain one.
//   g++ -shared -fPIC -O2 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/micro/src/memset_shim.cpp -o tools/micro/bin/memset_shim.so -ldl
//   LD_PRELOAD=tools/micro/bin/memset_shim.so UPP_MEMSET_LOG=/tmp/memsets.txt python ...
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <execinfo.h>
#include <cstdio>
#include <cstdlib>

static void *hip_sym(const char *name) {
    // (libamdhip64 arrives with a dlopen'ed Python extension: not in the scope RTLD_NEXT searches)
    static void *h = nullptr;
    if (!h) h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("/opt/rocm/lib/libamdhip64.so", RTLD_NOW);
    return h ? dlsym(h, name) : nullptr;
}
#include <fcntl.h>
#include <unistd.h>
#include <cstring>
static void note(const char *what, void *dst, size_t bytes, hipStream_t stream) {
    static auto is_capturing = (hipError_t(*)(hipStream_t, hipStreamCaptureStatus *))hip_sym("hipStreamIsCapturing");
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (!is_capturing || is_capturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusActive) return;
    const char *path = getenv("UPP_MEMSET_LOG");
    const int fd = open(path ? path : "/tmp/upp_memsets.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd < 0) return;
    char line[160];
    const int n = snprintf(line, sizeof line, "captured %s dst=%p bytes=%zu\n", what, dst, bytes);
    if (write(fd, line, n) < 0) {}
    void *bt[16];
    const int k = backtrace(bt, 16);
    backtrace_symbols_fd(bt, k, fd);
    close(fd);
}

extern "C" hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t stream) {
    static auto real = (hipError_t(*)(void *, int, size_t, hipStream_t))hip_sym("hipMemsetAsync");
    note("hipMemsetAsync", dst, bytes, stream);
    return real(dst, value, bytes, stream);
}
extern "C" hipError_t hipMemsetD8Async(hipDeviceptr_t dst, unsigned char value, size_t count, hipStream_t stream) {
    static auto real = (hipError_t(*)(hipDeviceptr_t, unsigned char, size_t, hipStream_t))hip_sym("hipMemsetD8Async");
    note("hipMemsetD8Async", (void *)dst, count, stream);
    return real(dst, value, count, stream);
}
extern "C" hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int value, size_t count, hipStream_t stream) {
    static auto real = (hipError_t(*)(hipDeviceptr_t, int, size_t, hipStream_t))hip_sym("hipMemsetD32Async");
    note("hipMemsetD32Async", (void *)dst, count * 4, stream);
    return real(dst, value, count, stream);
}
extern "C" hipError_t hipMemset2DAsync(void *dst, size_t pitch, int value, size_t width, size_t height, hipStream_t stream) {
    static auto real = (hipError_t(*)(void *, size_t, int, size_t, size_t, hipStream_t))hip_sym("hipMemset2DAsync");
    note("hipMemset2DAsync", dst, width * height, stream);
    return real(dst, pitch, value, width, height, stream);
}

// cross-stream edges while capturing: a wait that pulls a second stream into the capture is a FORK inside the graph (65-115 us per fork
// and join on this stack: NOTEBOOK 12.8); logged as "edge" lines
extern "C" hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t event, unsigned int flags) {
    static auto real = (hipError_t(*)(hipStream_t, hipEvent_t, unsigned int))hip_sym("hipStreamWaitEvent");
    static auto is_capturing = (hipError_t(*)(hipStream_t, hipStreamCaptureStatus *))hip_sym("hipStreamIsCapturing");
    hipStreamCaptureStatus before = hipStreamCaptureStatusNone, after = hipStreamCaptureStatusNone;
    if (is_capturing) is_capturing(stream, &before);
    const hipError_t rc = real(stream, event, flags);
    if (is_capturing) is_capturing(stream, &after);
    if (before == hipStreamCaptureStatusActive || after == hipStreamCaptureStatusActive) {
        const char *path = getenv("UPP_MEMSET_LOG");
        const int fd = open(path ? path : "/tmp/upp_memsets.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (fd >= 0) {
            char line[160];
            const int n = snprintf(line, sizeof line, "edge hipStreamWaitEvent stream=%p capturing before=%d after=%d\n", (void *)stream, (int)(before == hipStreamCaptureStatusActive),
                                   (int)(after == hipStreamCaptureStatusActive));
            if (write(fd, line, n) < 0) {}
            close(fd);
        }
    }
    return rc;
}
