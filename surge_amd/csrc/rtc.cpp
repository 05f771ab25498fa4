// rtc.cpp — run-time compilation of schema-specialised kernels (host side, no HIP runtime calls here).
//
// The ABI v2 slot fold is an interpreter when compiled ahead of time: slot types, operand sources and the operations a
// schema uses are kernel arguments.  A handle's schema never changes, so surge_replay_create_v2 compiles the same
// device source (fold_slots_device.h, embedded below byte for byte) once more with the schema as constants:
// hiprtc -> a gfx950 code object -> hipModuleLoadData (fold_slots.hip).  libhiprtc is dlopen'ed on first use — hosts
// without it keep the interpreter (surge_replay_kernel_info says which one a handle runs).
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>


// The device sources, embedded at build time (the assembler's .incbin finds them through -I surge_amd/csrc).  hipcc
// compiles this file as HIP too: the blobs belong to the host pass only.
#if defined(__HIP_DEVICE_COMPILE__)
#define SURGE_EMBED(sym, file) extern "C" const char sym[];
#else
#define SURGE_EMBED(sym, file)                                                                                         \
  __asm__(".pushsection .rodata\n.balign 16\n.hidden " #sym "\n.global " #sym "\n" #sym ":\n.incbin \"" file "\"\n.byte 0\n" \
          ".popsection\n");                                                                                            \
  extern "C" const char sym[];
#endif
SURGE_EMBED(surge_src_fold_layout_h, "fold_layout.h")
SURGE_EMBED(surge_src_fold_device_h, "fold_device.h")
SURGE_EMBED(surge_src_fold_slots_device_h, "fold_slots_device.h")
SURGE_EMBED(surge_src_fold_flat_device_h, "fold_flat_device.h")
SURGE_EMBED(surge_src_fold_chunk_device_h, "fold_chunk_device.h")
SURGE_EMBED(surge_src_fold_lane_device_h, "fold_lane_device.h")

namespace surge {
namespace {

typedef struct _hiprtcProgram* hiprtcProgram;
struct RtcApi {
  void* lib = nullptr;
  std::string path, why;
  int (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*CompileProgram)(hiprtcProgram, int, const char* const*) = nullptr;
  int (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
  int (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
  int (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
  int (*GetCode)(hiprtcProgram, char*) = nullptr;
  int (*DestroyProgram)(hiprtcProgram*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

std::mutex g_rtc_mu;
RtcApi g_rtc;
long g_rtc_cache_hits = 0;  // code objects served from the disk cache by this process
bool g_rtc_tried = false;

// libhiprtc: SURGE_HIPRTC_LIBRARY, then the copy that sits beside the HIP runtime this process already uses (PyTorch
// ships its own pair), then the loader's search path, then /opt/rocm.
bool load_rtc_locked() {
  if (g_rtc_tried) return g_rtc.lib != nullptr;
  g_rtc_tried = true;
  std::vector<std::string> cands;
  if (const char* v = std::getenv("SURGE_HIPRTC_LIBRARY")) cands.push_back(v);
  {
    Dl_info info;
    void* hip_fn = dlsym(RTLD_DEFAULT, "hipGetDeviceCount");
    if (hip_fn && dladdr(hip_fn, &info) && info.dli_fname) {
      std::string dir = info.dli_fname;
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        cands.push_back(dir.substr(0, slash + 1) + "libhiprtc.so");
        for (const char* v : {"7", "6"}) cands.push_back(dir.substr(0, slash + 1) + "libhiprtc.so." + v);
      }
    }
  }
  // runtime-only ROCm installs ship the versioned soname without the development symlink
  for (const char* n : {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6"}) cands.push_back(n);
  for (const char* n : {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6"}) cands.push_back(std::string("/opt/rocm/lib/") + n);
  for (const std::string& c : cands) {
    void* lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
      const char* e = dlerror();  // once: a second call returns NULL (the first one cleared the error)
      g_rtc.why += c + ": " + (e ? e : "?") + "; ";
      continue;
    }
    RtcApi a;
    a.lib = lib;
    a.path = c;
    bool ok = true;
    auto sym = [&](const char* n) -> void* {
      void* p = dlsym(lib, n);
      if (!p) ok = false;
      return p;
    };
    a.CreateProgram = (decltype(a.CreateProgram))sym("hiprtcCreateProgram");
    a.CompileProgram = (decltype(a.CompileProgram))sym("hiprtcCompileProgram");
    a.GetProgramLogSize = (decltype(a.GetProgramLogSize))sym("hiprtcGetProgramLogSize");
    a.GetProgramLog = (decltype(a.GetProgramLog))sym("hiprtcGetProgramLog");
    a.GetCodeSize = (decltype(a.GetCodeSize))sym("hiprtcGetCodeSize");
    a.GetCode = (decltype(a.GetCode))sym("hiprtcGetCode");
    a.DestroyProgram = (decltype(a.DestroyProgram))sym("hiprtcDestroyProgram");
    a.GetErrorString = (decltype(a.GetErrorString))sym("hiprtcGetErrorString");
    if (!ok) {
      g_rtc.why += c + ": missing hiprtc symbols; ";
      dlclose(lib);
      continue;
    }
    const std::string why = g_rtc.why;
    g_rtc = a;
    g_rtc.why = why;
    return true;
  }
  return false;
}

// ---- the code-object cache on disk ---------------------------------------------------------------------------------------
// A compile costs 0.6 - 1 s per schema and process; the result depends on nothing but the text compiled, the target and
// the compiler, so it is kept: <dir>/<128-bit key>.co, dir = $SURGE_REPLAY_CACHE_DIR, else $XDG_CACHE_HOME/surge_amd, else
// $HOME/.cache/surge_amd, else /tmp/surge_amd-<uid> (SURGE_REPLAY_CACHE=0: no cache).  A file is "SRGCO1\0\0" + length +
// FNV-1a of the code + the code; anything that does not check out is compiled again and overwritten (rename: atomic).
uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
  return h;
}

// The directory must be this user's own: a directory (not a link to one), owned by getuid(), writable by nobody else.
// Anything else — a directory someone else created first, a symbolic link, a group-writable path — and there is NO cache:
// a forged <key>.co would be loaded as GPU code without hiprtc ever running.  No fallback under /tmp either: a
// process without HOME / XDG_CACHE_HOME / SURGE_REPLAY_CACHE_DIR compiles every time.
bool own_private_dir(const std::string& d) {
  struct stat st;
  if (lstat(d.c_str(), &st) != 0) return false;
  return S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & 022) == 0;
}

std::string cache_dir() {
  if (const char* v = std::getenv("SURGE_REPLAY_CACHE"))
    if (std::atoi(v) == 0) return "";
  std::string d;
  if (const char* v = std::getenv("SURGE_REPLAY_CACHE_DIR")) d = v;
  else if (const char* x = std::getenv("XDG_CACHE_HOME")) d = std::string(x) + "/surge_amd";
  else if (const char* hme = std::getenv("HOME")) {
    const std::string c = std::string(hme) + "/.cache";
    (void)mkdir(c.c_str(), 0700);
    d = c + "/surge_amd";
  } else return "";
  (void)mkdir(d.c_str(), 0700);
  return own_private_dir(d) ? d : "";
}

std::string cache_key(const std::string& source, const char* const* headers, int n_headers, const char* const* opts, int n_opts) {
  uint64_t a = 1469598103934665603ull, b = 0x9E3779B97F4A7C15ull;
  auto mix = [&](const void* p, size_t n) {
    a = fnv1a(p, n, a);
    a = fnv1a("\x1f", 1, a);
    b = fnv1a(p, n, b ^ (uint64_t)n);
  };
  mix(source.data(), source.size());
  for (int i = 0; i < n_headers; ++i) mix(headers[i], std::strlen(headers[i]));
  for (int i = 0; i < n_opts; ++i) mix(opts[i], std::strlen(opts[i]));
  // the compiler: the HIP runtime's version (hiprtc ships with it) and, when it is loaded, hiprtc's own version and path — a
  // ROCm upgrade compiles afresh.  (Looked up in the libraries this process has loaded, whatever scope they were loaded with.)
  int ver = 0;
  void* f = dlsym(RTLD_DEFAULT, "hipRuntimeGetVersion");
  if (!f)
    for (const char* n : {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"})
      if (void* lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) {
        f = dlsym(lib, "hipRuntimeGetVersion");
        if (f) break;
      }
  if (f) (void)((int (*)(int*))f)(&ver);
  mix(&ver, sizeof ver);
  if (g_rtc.lib) {
    int major = 0, minor = 0;
    if (void* v = dlsym(g_rtc.lib, "hiprtcVersion")) (void)((int (*)(int*, int*))v)(&major, &minor);
    mix(&major, sizeof major);
    mix(&minor, sizeof minor);
    mix(g_rtc.path.data(), g_rtc.path.size());
  }
  if (ver == 0) return std::string();  // the compiler cannot be told apart from another one: nothing is cached
  char hex[40];
  std::snprintf(hex, sizeof hex, "%016llx%016llx", (unsigned long long)a, (unsigned long long)b);
  return hex;
}

bool cache_load(const std::string& path, std::vector<char>* code) {
  const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 022) != 0) {
    close(fd);
    return false;
  }
  FILE* f = fdopen(fd, "rb");
  if (!f) {
    close(fd);
    return false;
  }
  char head[24];
  bool ok = std::fread(head, 1, 24, f) == 24 && std::memcmp(head, "SRGCO1\0\0", 8) == 0;
  uint64_t len = 0, sum = 0;
  if (ok) {
    std::memcpy(&len, head + 8, 8);
    std::memcpy(&sum, head + 16, 8);
    ok = len > 0 && len < (1ull << 30);
  }
  if (ok) {
    code->resize((size_t)len);
    ok = std::fread(code->data(), 1, (size_t)len, f) == (size_t)len && std::fgetc(f) == EOF && fnv1a(code->data(), (size_t)len, 1469598103934665603ull) == sum;
  }
  std::fclose(f);
  if (!ok) code->clear();
  return ok;
}

void cache_store(const std::string& path, const std::vector<char>& code) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  (void)unlink(tmp.c_str());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  if (fd < 0) return;
  FILE* f = fdopen(fd, "wb");
  if (!f) {
    close(fd);
    return;
  }
  const uint64_t len = code.size(), sum = fnv1a(code.data(), code.size(), 1469598103934665603ull);
  bool ok = std::fwrite("SRGCO1\0\0", 1, 8, f) == 8 && std::fwrite(&len, 8, 1, f) == 1 && std::fwrite(&sum, 8, 1, f) == 1 && std::fwrite(code.data(), 1, code.size(), f) == code.size();
  ok = std::fclose(f) == 0 && ok;
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
}

}  // namespace

// Compile `source` (which #includes the embedded headers by their file names) for `arch`.  Returns false with *log set
// on any failure.  ms: wall time of the compilation.
bool rtc_compile(const std::string& source, const char* arch, std::vector<char>* code, std::string* log, double* ms) {
  std::lock_guard<std::mutex> lk(g_rtc_mu);
  const auto t0 = std::chrono::steady_clock::now();
  const char* headers[] = {surge_src_fold_layout_h, surge_src_fold_device_h, surge_src_fold_slots_device_h, surge_src_fold_flat_device_h,
                           surge_src_fold_chunk_device_h, surge_src_fold_lane_device_h};
  const char* names[] = {"fold_layout.h", "fold_device.h", "fold_slots_device.h", "fold_flat_device.h", "fold_chunk_device.h", "fold_lane_device.h"};
  constexpr int n_headers = 6;
  const std::string arch_opt = std::string("--offload-arch=") + arch;
  // -ffp-contract=off: an f64 ADD must round exactly like the JVM's (no fused multiply-add anywhere near it)
  const char* opts[] = {arch_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off"};
  const std::string dir = cache_dir();
  (void)load_rtc_locked();  // (before the key: the compiler's identity is part of it; a host without libhiprtc still hits entries made without it)
  const std::string key = cache_key(source, headers, n_headers, opts, 4);
  const std::string cached = (dir.empty() || key.empty()) ? "" : dir + "/" + key + ".co";
  if (const char* dump = std::getenv("SURGE_REPLAY_RTC_DUMP")) {  // the program text, for reading its ISA offline (hipcc -S -I surge_amd/csrc)
    if (FILE* f = std::fopen((std::string(dump) + "/" + key + ".hip").c_str(), "w")) {
      std::fwrite(source.data(), 1, source.size(), f);
      std::fclose(f);
    }
  }
  if (!cached.empty() && cache_load(cached, code)) {  // (needs no libhiprtc at all)
    g_rtc_cache_hits += 1;
    *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
  }
  if (!load_rtc_locked()) {
    *log = "libhiprtc not available: " + g_rtc.why;
    return false;
  }
  hiprtcProgram prog = nullptr;
  int rc = g_rtc.CreateProgram(&prog, source.c_str(), "surge_schema_spec.hip", n_headers, headers, names);
  if (rc != 0) {
    *log = std::string("hiprtcCreateProgram: ") + g_rtc.GetErrorString(rc);
    return false;
  }
  rc = g_rtc.CompileProgram(prog, 4, opts);
  size_t ls = 0;
  if (g_rtc.GetProgramLogSize(prog, &ls) == 0 && ls > 1) {
    log->assign(ls, '\0');
    g_rtc.GetProgramLog(prog, &(*log)[0]);
    while (!log->empty() && log->back() == '\0') log->pop_back();
  }
  bool ok = rc == 0;
  if (!ok) *log = std::string("hiprtcCompileProgram: ") + g_rtc.GetErrorString(rc) + "\n" + *log;
  if (ok) {
    size_t cs = 0;
    ok = g_rtc.GetCodeSize(prog, &cs) == 0 && cs > 0;
    if (ok) {
      code->resize(cs);
      ok = g_rtc.GetCode(prog, code->data()) == 0;
    }
    if (!ok) *log = "hiprtcGetCode failed";
  }
  g_rtc.DestroyProgram(&prog);
  if (ok && !cached.empty()) cache_store(cached, *code);
  *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return ok;
}

const char* rtc_library_path() {
  std::lock_guard<std::mutex> lk(g_rtc_mu);
  return g_rtc.lib ? g_rtc.path.c_str() : (g_rtc_cache_hits ? "hiprtc, earlier: the code-object cache on disk (no compile in this process)" : "");
}

}  // namespace surge
