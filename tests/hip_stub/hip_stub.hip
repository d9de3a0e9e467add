// A host-only stand-in for the HIP runtime, for ThreadSanitizer runs of the library's HOST threading on a machine
// without a GPU (tests/test_host_sanitizers.py; SURVEY.md section 5 "Race detection").  Test infrastructure only.
//   * a stream is an in-order worker thread executing queued closures;
//   * an event is a pair of counters (records issued / completed): hipStreamWaitEvent waits for the record issued
//     last before the call, as HIP does;
//   * device memory is host memory; copies and memsets run on the stream's thread;
//   * a kernel launch is a no-op -- except that a launch carrying a decoder State with a progress word publishes
//     (epoch, iteration, codewords running) there as the real kernels do, with "running" scripted by HIP_STUB_DONE_AT
//     (0 from that iteration on: the host's early-termination paths run);
//   * HIP_STUB_FAIL=<function>:<n> makes the n-th call of that function return hipErrorUnknown.
// Built with the same host-only clang mode as the library objects it is linked with (it needs dev::State).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

// (the kernel header defines non-template kernels: a private copy of its namespace keeps the stub from clashing with
// the library objects it is linked with; only dev::State and dev::progress_word are used here)
#define dev dev_for_the_stub
#include "../../ldpc_toolbox_amd/csrc/kernels.hip.h"
#undef dev

namespace {

struct Stream {
  std::thread worker;
  std::mutex m;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool closing = false;
  uint64_t submitted = 0, completed = 0;
  Stream() {
    worker = std::thread([this] {
      for (;;) {
        std::function<void()> f;
        {
          std::unique_lock<std::mutex> lock(m);
          cv.wait(lock, [&] { return closing || !q.empty(); });
          if (q.empty()) return;
          f = std::move(q.front());
          q.pop_front();
        }
        f();
        {
          std::lock_guard<std::mutex> lock(m);
          completed++;
        }
        idle.notify_all();
      }
    });
  }
  void push(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lock(m);
      q.push_back(std::move(f));
      submitted++;
    }
    cv.notify_one();
  }
  void sync() {
    std::unique_lock<std::mutex> lock(m);
    idle.wait(lock, [&] { return completed == submitted; });
  }
  bool busy() {
    std::lock_guard<std::mutex> lock(m);
    return completed != submitted;
  }
  ~Stream() {
    {
      std::lock_guard<std::mutex> lock(m);
      closing = true;
    }
    cv.notify_all();
    if (worker.joinable()) worker.join();
  }
};

struct EventState {
  std::mutex m;
  std::condition_variable cv;
  uint64_t issued = 0, completed = 0;
};
// (the handle owns a reference, and so does every queued closure: destroying an event that streams still refer to is
// legal in HIP)
struct Event {
  std::shared_ptr<EventState> st = std::make_shared<EventState>();
};

Stream *default_stream() {
  static Stream *s = new Stream();  // (leaked on purpose: outlives every static destructor)
  return s;
}
Stream *as_stream(hipStream_t s) { return s ? reinterpret_cast<Stream *>(s) : default_stream(); }

std::mutex g_reg_mutex;
std::map<const void *, std::string> &kernel_names() {
  static std::map<const void *, std::string> m;
  return m;
}

struct LaunchConfig {
  dim3 grid, block;
  size_t shmem;
  hipStream_t stream;
};
thread_local LaunchConfig t_config;

std::atomic<uint64_t> g_launches{0};

bool fail_now(const char *fn) {
  static const char *spec = std::getenv("HIP_STUB_FAIL");
  if (!spec) return false;
  static std::mutex m;
  static std::map<std::string, uint64_t> calls;
  const char *colon = std::strchr(spec, ':');
  if (!colon || std::strncmp(spec, fn, size_t(colon - spec)) != 0 || std::strlen(fn) != size_t(colon - spec)) return false;
  std::lock_guard<std::mutex> lock(m);
  return ++calls[fn] == std::strtoull(colon + 1, nullptr, 10);
}
#define STUB_FAIL(fn) \
  if (fail_now(fn)) return hipErrorUnknown

// index of the dev::State argument of the kernels that publish the progress word, -1 for the others
int state_arg(const std::string &name) {
  for (const char *k : {"cn_minsum_kernel", "cn_minsum_lfree_kernel", "cn_minsum_rec_kernel", "cn_staged_kernel", "hl_level_kernel",
                        "hl_level_reg_kernel", "hl_minsum_kernel", "hl_minsum_reg_kernel", "hl_minsum_rec_kernel", "cn_i8_kernel",
                        "hl_i8_kernel", "hl_i8_reg_kernel"})
    if (name.find(k) != std::string::npos) return 2;  // (Graph, Sched, State, ...)
  if (name.find("hl_slice_kernel") != std::string::npos) return 1;  // (Graph, State, ...)
  if (name.find("compact_plan_kernel") != std::string::npos) return 0;
  return -1;
}

}  // namespace

extern "C" {

hipError_t hipGetDeviceCount(int *count) {
  *count = 1;
  return hipSuccess;
}
hipError_t hipSetDevice(int) {
  STUB_FAIL("hipSetDevice");
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *value, hipDeviceAttribute_t attr, int) {
  *value = attr == hipDeviceAttributeMultiprocessorCount ? 256 : 0;
  return hipSuccess;
}
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *blocks, const void *, int, size_t) {
  *blocks = 1;
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
std::atomic<int> g_last_error{0};  // a failed kernel launch is reported by the next hipGetLastError, as in HIP
hipError_t hipGetLastError(void) { return static_cast<hipError_t>(g_last_error.exchange(0)); }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "injected by the HIP stub"; }

hipError_t hipMalloc(void **p, size_t bytes) {
  STUB_FAIL("hipMalloc");
  *p = std::calloc(std::max<size_t>(bytes, 1), 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned int) {
  *p = std::calloc(std::max<size_t>(bytes, 1), 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) {
  std::free(p);
  return hipSuccess;
}
// (every pointer of the stand-in runtime is host memory: Simulator::generate then takes its copy-out path)
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *attr, const void *) {
  std::memset(attr, 0, sizeof(*attr));
  attr->type = hipMemoryTypeHost;
  return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned int) {
  *dev = host;
  return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) {
  *s = reinterpret_cast<hipStream_t>(new Stream());
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  delete reinterpret_cast<Stream *>(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  as_stream(s)->sync();
  STUB_FAIL("hipStreamSynchronize");
  return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) { return as_stream(s)->busy() ? hipErrorNotReady : hipSuccess; }

hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int) {
  *e = reinterpret_cast<hipEvent_t>(new Event());
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
  delete reinterpret_cast<Event *>(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t s) {
  STUB_FAIL("hipEventRecord");
  std::shared_ptr<EventState> e = reinterpret_cast<Event *>(ev)->st;
  uint64_t gen;
  {
    std::lock_guard<std::mutex> lock(e->m);
    gen = ++e->issued;
  }
  as_stream(s)->push([e, gen] {
    {
      std::lock_guard<std::mutex> lock(e->m);
      e->completed = std::max(e->completed, gen);
    }
    e->cv.notify_all();
  });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t ev, unsigned int) {
  STUB_FAIL("hipStreamWaitEvent");
  std::shared_ptr<EventState> e = reinterpret_cast<Event *>(ev)->st;
  uint64_t target;
  {
    std::lock_guard<std::mutex> lock(e->m);
    target = e->issued;  // the record made last before this call (none: the wait is a no-op)
  }
  as_stream(s)->push([e, target] {
    std::unique_lock<std::mutex> lock(e->m);
    e->cv.wait(lock, [&] { return e->completed >= target; });
  });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t ev) {
  std::shared_ptr<EventState> e = reinterpret_cast<Event *>(ev)->st;
  std::unique_lock<std::mutex> lock(e->m);
  e->cv.wait(lock, [&] { return e->completed >= e->issued; });
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
  *ms = 0.01f;
  return hipSuccess;
}

hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t s) {
  STUB_FAIL("hipMemcpyAsync");
  as_stream(s)->push([=] { std::memcpy(dst, src, bytes); });
  return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) {
  default_stream()->sync();
  std::memcpy(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s) {
  as_stream(s)->push([=] { std::memset(dst, value, bytes); });
  return hipSuccess;
}

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
  t_config = LaunchConfig{grid, block, shmem, stream};
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream) {
  *grid = t_config.grid;
  *block = t_config.block;
  *shmem = t_config.shmem;
  *stream = t_config.stream;
  return hipSuccess;
}
void **__hipRegisterFatBinary(const void *) {
  static void *handle = nullptr;
  return &handle;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *host_fun, char *, const char *device_name, unsigned int, void *, void *, void *,
                           void *, int *) {
  std::lock_guard<std::mutex> lock(g_reg_mutex);
  kernel_names()[host_fun] = device_name;
}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}

hipError_t hipLaunchKernel(const void *fun, dim3, dim3, void **args, size_t, hipStream_t s) {
  if (fail_now("hipLaunchKernel")) {
    g_last_error = static_cast<int>(hipErrorUnknown);
    return hipErrorUnknown;
  }
  g_launches++;
  int at = -1;
  {
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    auto it = kernel_names().find(fun);
    if (it != kernel_names().end()) at = state_arg(it->second);
  }
  if (at < 0) {
    as_stream(s)->push([] {});
    return hipSuccess;
  }
  const ldpc::dev_for_the_stub::State st = *static_cast<const ldpc::dev_for_the_stub::State *>(args[at]);  // (args live on the caller's stack)
  static const unsigned long done_at = std::getenv("HIP_STUB_DONE_AT") ? std::strtoul(std::getenv("HIP_STUB_DONE_AT"), nullptr, 10) : ~0ul;
  as_stream(s)->push([st] {
    if (st.publish != nullptr) {
      const uint32_t running = st.tick >= done_at ? 0u : 1u;
      __atomic_store_n(st.publish, ldpc::dev_for_the_stub::progress_word(st.epoch, st.tick, running), __ATOMIC_RELEASE);
    }
  });
  return hipSuccess;
}

unsigned long long hip_stub_launches(void) { return g_launches.load(); }

}  // extern "C"
