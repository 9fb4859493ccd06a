// altro_capi.cpp — extern "C" front end of libaltro_hip.so (see include/altro_hip.h).
//
// Records the problem definition exactly as the reference's altro::problem::Problem setters do,
// creates the device engine lazily on the first compute call, and forwards every entry point.
// No exception crosses the boundary; there is NO CPU fallback.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <new>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "altro_common.hpp"

using namespace altro_hip;

extern "C" int altro_chain_claim(int device, int delta);

#define ALTRO_USER_PLUGIN_ABI_HOST 7  // must equal ALTRO_USER_PLUGIN_ABI of altro_user_model.hpp

struct altro_solver_s {
  ProblemSpec spec;
  altro_options opts;
  std::unique_ptr<EngineBase> engine;
  bool uploaded = false;
  bool ilqr_mode = false;
  std::string err;
  // Asynchronous solve (altro_solve_al_async): ONE worker thread per handle, created on first use and
  // parked on a condition variable between solves.  While a solve is pending the handle only answers
  // altro_solve_poll / altro_wait / altro_last_error: everything else returns ALTRO_NOT_READY, so a caller
  // cannot change options, inputs or device buffers under the solve in flight.
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  bool job_posted = false, worker_exit = false;
  std::atomic<int> async_done{1};
  bool async_pending = false;
  altro_status async_status = ALTRO_OK;
  std::string async_err;  // the worker's own error text, copied to err by altro_wait
  ~altro_solver_s() {
    if (worker.joinable()) {
      {
        std::lock_guard<std::mutex> lk(mu);
        worker_exit = true;
      }
      cv.notify_all();
      worker.join();
    }
  }
};

static thread_local std::string g_create_error;

// ---- user-model plugins (altro_register_model_source; see altro_user_model.hpp) ---------------------------------
namespace {
struct UserModelEntry {
  std::string name, so_path;
  void* dl = nullptr;
  int n = 0, m = 0;
  EngineBase* (*make)(const altro_desc*, std::string*) = nullptr;
  int (*check)(int, const double*, int, double, double*) = nullptr;
  int (*check_functors)(int, const double*, int, double, double*, int*) = nullptr;
  int functors = 0;  // bit 0: the source defines cost types, bit 1: constraint types
  int cost_types = 0, con_types = 0;
  bool checked = false;
  uint64_t hash = 0;
};
std::mutex g_models_mu;
std::vector<UserModelEntry> g_models;  // kind = ALTRO_MODEL_USER_BASE + index

uint64_t Fnv1a(const std::string& s, uint64_t h = 1469598103934665603ULL) {
  for (unsigned char ch : s) {
    h ^= ch;
    h *= 1099511628211ULL;
  }
  return h;
}
bool ReadFile(const std::string& path, std::string* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}
std::string Tail(const std::string& s, size_t n) { return s.size() > n ? s.substr(s.size() - n) : s; }

// Runs argv[0] with the given arguments, stdout and stderr into `log`: posix_spawn with an argument vector -- no shell,
// so no quoting of paths, and independent of the host application's SIGCHLD disposition as far as waitpid allows
// (std::system() returns -1 when SIGCHLD is ignored).  Returns the exit code, or -1.
extern "C" char** environ;
int RunTool(const std::vector<std::string>& args, const std::string& log) {
  posix_spawn_file_actions_t fa;
  if (posix_spawn_file_actions_init(&fa) != 0) return -1;
  posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
  posix_spawn_file_actions_adddup2(&fa, 1, 2);
  std::vector<char*> argv;
  for (const std::string& a : args) argv.push_back(const_cast<char*>(a.c_str()));
  argv.push_back(nullptr);
  pid_t pid = 0;
  const int rc = posix_spawn(&pid, argv[0], &fa, nullptr, argv.data(), environ);
  posix_spawn_file_actions_destroy(&fa);
  if (rc != 0) return -1;
  int status = 0;
  for (;;) {
    const pid_t w = waitpid(pid, &status, 0);
    if (w == pid) break;
    if (w < 0 && errno != EINTR) return -1;  // (ECHILD: the host reaps children itself -- the result file decides)
  }
  return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}
// [A-Za-z0-9_] only: the name goes into the generated source (a comment and a #line directive)
std::string SafeName(const char* name) {
  std::string out;
  for (const char* p = name; *p && out.size() < 64; ++p)
    out += (std::isalnum(static_cast<unsigned char>(*p)) || *p == '_') ? *p : '_';
  return out.empty() ? std::string("model") : out;
}

// FunctionBase::CheckJacobian (functionbase.cpp:35-73) on the device: 64 points uniform in [-1, 1]^(n+m)
// (VectorXd::Random), central differences with 1e-6, tolerance kDefaultTolerance = 1e-4 (functionbase.hpp:86) on every
// ENTRY's error relative to max(1, |J_ij|): the reference's forward-difference helper is an opt-in test utility, this
// check gates registration and must not reject a correct Jacobian of a model with large second derivatives -- nor let a
// wrong entry of order one hide behind a large norm of the whole matrix
altro_status CheckUserJacobian(UserModelEntry& e, int device, std::string* err) {
  if (e.checked) return ALTRO_OK;
  const int samples = 64, nm = e.n + e.m;
  std::mt19937_64 gen(e.hash);
  std::vector<double> z((size_t)samples * nm);
  for (double& v : z) v = -1.0 + 2.0 * (static_cast<double>(gen() >> 11) * 0x1p-53);
  double max_err = 0.0;
  const int rc = e.check(device, z.data(), samples, 1e-6, &max_err);
  if (rc != 0) {
    *err = "user model '" + e.name + "': the Jacobian check could not run on the device (code " + std::to_string(rc) + ")";
    return ALTRO_HIP_ERROR;
  }
  if (!(max_err < 1e-4)) {
    char buf[256];
    snprintf(buf, sizeof(buf), "user model '%s': jac() does not match finite differences of f(): max_ij |J_fd - J|_ij / max(1, |J_ij|) = %.3g >= 1e-4 "
             "(FunctionBase::CheckJacobian)", e.name.c_str(), max_err);
    *err = buf;
    return ALTRO_INVALID_ARG;
  }
  if (e.functors) {
    // ScalarFunction::CheckGradient, FunctionBase::CheckHessian / CheckJacobian (functionbase.cpp:42-125) for the
    // user's cost and constraint, at the same points
    double errs[3] = {0, 0, 0};
    int worst[3] = {0, 0, 0};
    const int rc2 = e.check_functors(device, z.data(), samples, 1e-6, errs, worst);
    if (rc2 != 0) {
      *err = "user model '" + e.name + "': the cost / constraint derivative check could not run on the device (code " +
             std::to_string(rc2) + ")";
      return ALTRO_HIP_ERROR;
    }
    const char* what[3] = {"UserCost::gradient() does not match finite differences of eval()",
                           "UserCost::hessian() does not match finite differences of gradient()",
                           "UserConstraint::jacobian() does not match finite differences of eval()"};
    for (int q = 0; q < 3; ++q)
      if (!(errs[q] < 1e-4)) {
        char buf[360];
        snprintf(buf, sizeof(buf), "user model '%s': %s (%s type %d): error %.3g >= 1e-4 (FunctionBase::Check%s)", e.name.c_str(),
                 what[q], q < 2 ? "cost" : "constraint", worst[q], errs[q], q == 0 ? "Gradient" : q == 1 ? "Hessian" : "Jacobian");
        *err = buf;
        return ALTRO_INVALID_ARG;
      }
  }
  e.checked = true;
  return ALTRO_OK;
}
}  // namespace

extern "C" {

void altro_default_options(altro_options* o) {  // altro/common/solver_options.hpp:23-56
  std::memset(o, 0, sizeof(*o));
  o->max_iterations_total = 300;
  o->max_iterations_outer = 30;
  o->max_iterations_inner = 100;
  o->cost_tolerance = 1e-4;
  o->gradient_tolerance = 1e-2;
  o->bp_reg_increase_factor = 1.6;
  o->bp_reg_enable = 1;
  o->bp_reg_initial = 0.0;
  o->bp_reg_max = 1e8;
  o->bp_reg_min = 1e-8;
  o->bp_reg_fail_threshold = 100;
  o->check_forwardpass_bounds = 1;
  o->state_max = 1e8;
  o->control_max = 1e8;
  o->line_search_max_iterations = kLineSearchLanes < 20 ? kLineSearchLanes : 20;  // (20; fewer only in experimental ALTRO_LS_LANES builds)
  o->line_search_lower_bound = 1e-8;
  o->line_search_upper_bound = 10.0;
  o->line_search_decrease_factor = 2;
  o->constraint_tolerance = 1e-4;
  o->maximum_penalty = 1e8;
  o->initial_penalty = 1.0;
  o->reset_duals = 1;
  o->profiler_enable = 0;
}

altro_status altro_create(const altro_desc* desc, altro_handle* out) {
  if (!desc || !out) return ALTRO_INVALID_ARG;
  if (desc->n <= 0 || desc->m <= 0 || desc->N <= 0 || desc->batch <= 0 ||
      (desc->dtype != ALTRO_F64 && desc->dtype != ALTRO_F32)) {
    g_create_error = "invalid altro_desc";
    return ALTRO_INVALID_ARG;
  }
  altro_solver_s* h = new (std::nothrow) altro_solver_s();
  if (!h) return ALTRO_INVALID_ARG;
  h->spec.desc = *desc;
  altro_default_options(&h->opts);
  *out = h;
  return ALTRO_OK;
}

void altro_destroy(altro_handle h) { delete h; }

altro_status altro_get_desc(altro_handle h, altro_desc* out) {
  if (!h || !out) return ALTRO_INVALID_ARG;
  *out = h->spec.desc;
  return ALTRO_OK;
}

const char* altro_last_error(altro_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

}  // extern "C"

namespace {

// Create the engine (needs the model) and upload the recorded problem.
altro_status Ensure(altro_handle h) {
  if (!h) return ALTRO_INVALID_ARG;
  if (h->uploaded) return ALTRO_OK;
  const altro_desc& d = h->spec.desc;
  if (!h->engine) {
    EngineBase* e = nullptr;
    std::string err = "unsupported (model, n, m, dtype) combination";
    const bool f64 = d.dtype == ALTRO_F64;
    switch (h->spec.model_kind) {
      case ALTRO_MODEL_UNICYCLE:
        e = f64 ? MakeEngineUnicycleF64(d, &err) : MakeEngineUnicycleF32(d, &err);
        break;
      case ALTRO_MODEL_TRIPLE_INTEGRATOR:
        if (h->spec.dof == 2) e = f64 ? MakeEngineTripleInt2F64(d, &err) : MakeEngineTripleInt2F32(d, &err);
        break;
      case ALTRO_MODEL_QUADROTOR12:
        e = f64 ? MakeEngineQuad12F64(d, &err) : MakeEngineQuad12F32(d, &err);
        break;
      default:
        if (h->spec.model_kind >= ALTRO_MODEL_USER_BASE) {
          std::lock_guard<std::mutex> lk(g_models_mu);
          const size_t idx = (size_t)(h->spec.model_kind - ALTRO_MODEL_USER_BASE);
          if (idx >= g_models.size()) {
            err = "unknown user model kind (altro_register_model_source returns the kind)";
            break;
          }
          UserModelEntry& um = g_models[idx];
          if (um.n != d.n || um.m != d.m) {
            err = "state/control dimensions do not match the user model";
            break;
          }
          // the Jacobian check runs once per model, at registration or -- registered without a device -- here
          if (CheckUserJacobian(um, d.device_id, &err) != ALTRO_OK) break;
          e = um.make(&d, &err);
        } else {
          err = "altro_set_model has not been called";
        }
        break;
    }
    if (!e) {
      h->err = err;
      return (err.find("hip") != std::string::npos || err.find(".hpp:") != std::string::npos) ? ALTRO_HIP_ERROR
                                                                                                : ALTRO_UNSUPPORTED;
    }
    h->engine.reset(e);
  }
  altro_status st = h->engine->Upload(h->spec, &h->err);
  if (st == ALTRO_OK) h->uploaded = true;
  return st;
}

// true (and the error text set) while an asynchronous solve owns the handle
bool Busy(altro_handle h) {
  if (!h->async_pending) return false;
  h->err = "an asynchronous solve is pending on this handle (call altro_wait first)";
  return true;
}

template <class F>
altro_status Forward(altro_handle h, F f) {
  if (h && Busy(h)) return ALTRO_NOT_READY;
  altro_status st = Ensure(h);
  if (st != ALTRO_OK) return st;
  st = f(*h->engine);
  if (st != ALTRO_OK) h->err = h->engine->LastError();
  return st;
}

altro_status DefChanged(altro_handle h) {
  if (h->uploaded) {
    h->err = "the problem definition cannot change after the first compute call; create a new handle";
    return ALTRO_NOT_READY;
  }
  return ALTRO_OK;
}

}  // namespace

extern "C" {

altro_status altro_register_model_source(const char* name, const char* source, int check_jacobian, int* kind_out) {
  if (!name || !source || !kind_out) return ALTRO_INVALID_ARG;
  std::string& err = g_create_error;
  // the engine headers live next to the library (in-tree: altro-cpp_amd/csrc); ALTRO_HIP_INCLUDE_DIR overrides
  std::string inc;
  if (const char* e = std::getenv("ALTRO_HIP_INCLUDE_DIR")) {
    inc = e;
  } else {
    Dl_info info;
    if (!dladdr(reinterpret_cast<void*>(&altro_create), &info) || !info.dli_fname) {
      err = "cannot locate libaltro_hip.so (dladdr)";
      return ALTRO_HIP_ERROR;
    }
    inc = info.dli_fname;
    const size_t slash = inc.find_last_of('/');
    inc = slash == std::string::npos ? "." : inc.substr(0, slash);
  }
  std::string hdrs;
  for (const char* f : {"altro_common.hpp", "altro_device.hpp", "altro_kernels.hpp", "altro_engine.hpp", "altro_user_model.hpp",
                        "../../include/altro_hip.h"}) {
    std::string txt;
    if (!ReadFile(inc + "/" + f, &txt)) {
      err = std::string("engine header ") + f + " not found in " + inc + " (set ALTRO_HIP_INCLUDE_DIR)";
      return ALTRO_NOT_READY;
    }
    hdrs += txt;
  }
  std::string arch = "gfx950";
  if (const char* e = std::getenv("ALTRO_HIP_ARCH")) {
    arch = e;
  } else {
    hipDeviceProp_t p;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipGetDeviceProperties(&p, 0) == hipSuccess) {
      arch = p.gcnArchName;
      arch = arch.substr(0, arch.find(':'));
    }
  }
  const std::vector<std::string> flagv = {"-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast", "-mllvm",
                                          "-amdgpu-mfma-vgpr-form=1", "-Wno-unused-value", "-Wno-unused-result",
                                          "--offload-arch=" + arch};
  std::string flags;
  for (const std::string& f : flagv) flags += f + " ";
  const uint64_t hash = Fnv1a(flags, Fnv1a(hdrs, Fnv1a(source)));
  std::lock_guard<std::mutex> lk(g_models_mu);
  for (size_t i = 0; i < g_models.size(); ++i)
    if (g_models[i].hash == hash) {  // same source, same library: already loaded
      *kind_out = ALTRO_MODEL_USER_BASE + (int)i;
      return ALTRO_OK;
    }
  std::string cache = inc + "/_user_cache";
  if (const char* e = std::getenv("ALTRO_HIP_CACHE_DIR")) cache = e;
  mkdir(cache.c_str(), 0700);  // (plugins are code: private to the user.  Stale entries -- every header edit changes the
                               //  hash -- are the deployment's to prune: altro_user_model_path names the live ones)
  const std::string safe = SafeName(name);
  char hx[32];
  snprintf(hx, sizeof(hx), "%016llx", (unsigned long long)hash);
  const std::string stem = cache + "/altro_user_" + hx;
  const std::string so = stem + ".so";
  // A cached plugin is trusted only if it says itself that it was built from this very (source, headers, flags): the
  // generated translation unit embeds the hash, checked after dlopen.  A file that merely carries the right name is
  // recompiled.
  auto embedded_hash_ok = [&](void* dl) {
    auto fn = reinterpret_cast<unsigned long long (*)()>(dlsym(dl, "altro_user_source_hash"));
    return fn && fn() == (unsigned long long)hash;
  };
  void* dl = nullptr;
  if (access(so.c_str(), R_OK) == 0) {
    dl = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (dl && !embedded_hash_ok(dl)) {
      dlclose(dl);
      dl = nullptr;
    }
    if (!dl) unlink(so.c_str());
  }
  if (!dl) {
    const std::string src = stem + ".hip", log = stem + ".log", tmp = stem + "." + std::to_string((long)getpid()) + ".tmp.so";
    {
      std::ofstream f(src);
      f << "// generated by altro_register_model_source for the user model '" << safe << "'\n"
        << "#include <hip/hip_runtime.h>\n#define ALTRO_MODEL_FN __device__ __forceinline__\n"
        // (the user's functions under the language-level contraction rule, like the generic integrators that call them:
        //  altro_device.hpp, "CONTRACTION MODE OF THE GENERIC DYNAMICS CODE")
        << "#pragma clang fp contract(on)\n"
        << "namespace altro_user {\n#line 1 \"user model " << safe << "\"\n" << source << "\n}  // namespace altro_user\n"
        << "#pragma clang fp contract(fast)\n"
        << "#include \"altro_user_model.hpp\"  // (the engine headers see ALTRO_USER_COST / ALTRO_USER_CONSTRAINT)\n"
        << "extern \"C\" unsigned long long altro_user_source_hash() { return 0x" << hx << "ULL; }\n";
      if (!f) {
        err = "cannot write " + src + " (set ALTRO_HIP_CACHE_DIR to a writable directory)";
        return ALTRO_NOT_READY;
      }
    }
    std::string hipcc = "/opt/rocm/bin/hipcc";
    if (const char* e = std::getenv("HIPCC")) hipcc = e;
    std::vector<std::string> args = {hipcc};
    args.insert(args.end(), flagv.begin(), flagv.end());
    args.push_back("-I" + inc);
    args.push_back("-o");
    args.push_back(tmp);
    args.push_back(src);
    const int rc = RunTool(args, log);
    // (rc == -1 with a result file: waitpid failed because the host application reaps its children itself -- the exit
    //  code is unknown, so the file must prove itself before it gets the cached name: it loads and carries this hash)
    bool tmp_ok = rc == 0;
    if (rc == -1 && access(tmp.c_str(), R_OK) == 0) {
      if (void* probe = dlopen(tmp.c_str(), RTLD_NOW | RTLD_LOCAL)) {
        tmp_ok = embedded_hash_ok(probe);
        dlclose(probe);
      }
    }
    if (!tmp_ok || rename(tmp.c_str(), so.c_str()) != 0) {
      std::string out;
      ReadFile(log, &out);
      err = "compiling the user model '" + safe + "' failed:\n" + Tail(out, 1500);
      unlink(tmp.c_str());
      return ALTRO_INVALID_ARG;
    }
    dl = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
  }
  UserModelEntry e;
  e.name = safe;
  e.so_path = so;
  e.hash = hash;
  e.dl = dl;
  if (!e.dl) {
    err = std::string("dlopen of the user-model plugin failed: ") + dlerror();
    return ALTRO_HIP_ERROR;
  }
  auto abi = reinterpret_cast<int (*)()>(dlsym(e.dl, "altro_user_abi"));
  auto dims = reinterpret_cast<void (*)(int*, int*)>(dlsym(e.dl, "altro_user_dims"));
  e.make = reinterpret_cast<EngineBase* (*)(const altro_desc*, std::string*)>(dlsym(e.dl, "altro_user_make_engine"));
  e.check = reinterpret_cast<int (*)(int, const double*, int, double, double*)>(dlsym(e.dl, "altro_user_check_jacobian"));
  auto finfo = reinterpret_cast<int (*)(int*, int*)>(dlsym(e.dl, "altro_user_functor_info"));
  e.check_functors =
      reinterpret_cast<int (*)(int, const double*, int, double, double*, int*)>(dlsym(e.dl, "altro_user_check_functors"));
  if (!abi || !dims || !e.make || !e.check || !finfo || !e.check_functors || abi() != ALTRO_USER_PLUGIN_ABI_HOST ||
      !embedded_hash_ok(e.dl)) {
    err = "the cached user-model plugin " + so + " does not match this library; delete it";
    dlclose(e.dl);
    return ALTRO_HIP_ERROR;
  }
  // one book of chained engines for built-in and plugin engines (altro_engine.hpp: ChainClaim)
  if (auto hook = reinterpret_cast<void (*)(int (*)(int, int))>(dlsym(e.dl, "altro_user_set_chain_hook"))) hook(&altro_chain_claim);
  dims(&e.n, &e.m);
  e.functors = finfo(&e.cost_types, &e.con_types);
  if (check_jacobian) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {  // without a device the check runs at first use
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess) dev = 0;  // clear a sticky "no device selected" state; device 0 is fine
      altro_status st = CheckUserJacobian(e, dev, &err);
      if (st != ALTRO_OK) {
        dlclose(e.dl);
        return st;
      }
    }
  } else {
    e.checked = true;  // the caller vouches for the Jacobian
  }
  g_models.push_back(e);
  *kind_out = ALTRO_MODEL_USER_BASE + (int)g_models.size() - 1;
  return ALTRO_OK;
}

// Engines with chains of sweeps per device: the counter every engine of the process books in -- the built-in ones through
// this library's copy of the counter (altro_common.hpp: ChainClaimLocal), plugins through the hook they are handed at load.
int altro_chain_claim(int device, int delta) { return altro_hip::ChainClaimLocal(device, delta); }

int altro_user_model_path(int kind, char* buf, int len) {
  std::lock_guard<std::mutex> lk(g_models_mu);
  const int idx = kind - ALTRO_MODEL_USER_BASE;
  if (idx < 0 || idx >= (int)g_models.size() || !buf || len <= 0) return -1;
  std::snprintf(buf, (size_t)len, "%s", g_models[idx].so_path.c_str());
  return (int)g_models[idx].so_path.size();
}

altro_status altro_set_model(altro_handle h, int kind, const double* params, int nparams) {
  if (!h) return ALTRO_INVALID_ARG;
  if (DefChanged(h) != ALTRO_OK) return ALTRO_NOT_READY;
  h->spec.model_kind = kind;
  h->spec.dof = (kind == ALTRO_MODEL_TRIPLE_INTEGRATOR && params && nparams > 0) ? (int)params[0] : 0;
  return ALTRO_OK;
}
altro_status altro_set_knot_models(altro_handle h, const int* model_of_knot, int count) {
  if (!h || !model_of_knot) return ALTRO_INVALID_ARG;
  if (DefChanged(h) != ALTRO_OK) return ALTRO_NOT_READY;
  if (count != h->spec.desc.N) {
    h->err = "altro_set_knot_models: expected N = " + std::to_string(h->spec.desc.N) + " indices (the terminal knot has no dynamics)";
    return ALTRO_INVALID_ARG;
  }
  for (int k = 0; k < count; ++k)
    if (model_of_knot[k] < 0) {
      h->err = "altro_set_knot_models: index " + std::to_string(k) + " is negative";
      return ALTRO_INVALID_ARG;
    }
  h->spec.knot_model.assign(model_of_knot, model_of_knot + count);  // (checked against the model's list at the upload)
  return ALTRO_OK;
}
altro_status altro_set_uniform_step(altro_handle h, float hstep) {
  if (!h || !(hstep > 0.0f)) return ALTRO_INVALID_ARG;
  if (Busy(h)) return ALTRO_NOT_READY;
  // the step belongs to the trajectory (trajectory.hpp:122-130), not to the problem definition: it may
  // change between solves, also after the device state exists
  h->spec.hstep = hstep;
  h->spec.hk.clear();  // SetUniformStep overwrites every knot's step and time (trajectory.hpp:122-130)
  h->spec.tk.clear();
  if (h->uploaded) {
    altro_status st = h->engine->SetStep(hstep);
    if (st == ALTRO_OK) st = h->engine->SetKnotTimes(h->spec, &h->err);
    else h->err = h->engine->LastError();
    return st;
  }
  return ALTRO_OK;
}
altro_status altro_set_steps(altro_handle h, const float* hk, int count) {
  if (!h || !hk) return ALTRO_INVALID_ARG;
  if (Busy(h)) return ALTRO_NOT_READY;
  if (count != h->spec.desc.N) {
    h->err = "altro_set_steps: expected N = " + std::to_string(h->spec.desc.N) + " steps (the terminal knot has none)";
    return ALTRO_INVALID_ARG;
  }
  for (int k = 0; k < count; ++k)
    if (!(hk[k] > 0.0f)) {
      h->err = "altro_set_steps: step " + std::to_string(k) + " is not positive";
      return ALTRO_INVALID_ARG;
    }
  if (h->spec.tk.empty() && h->spec.hk.empty() && h->spec.hstep > 0.0f) {
    // the times an earlier SetUniformStep left stay (Trajectory::SetStep does not touch them)
    const int N = h->spec.desc.N;
    h->spec.tk.resize(N + 1);
    for (int k = 0; k < N; ++k) h->spec.tk[k] = static_cast<float>(k) * h->spec.hstep;
    h->spec.tk[N] = h->spec.hstep * N;
  }
  h->spec.hk.assign(hk, hk + count);
  if (h->uploaded) return h->engine->SetKnotTimes(h->spec, &h->err);
  return ALTRO_OK;
}
altro_status altro_set_times(altro_handle h, const float* tk, int count) {
  if (!h || !tk) return ALTRO_INVALID_ARG;
  if (Busy(h)) return ALTRO_NOT_READY;
  if (count != h->spec.desc.N + 1) {
    h->err = "altro_set_times: expected N + 1 = " + std::to_string(h->spec.desc.N + 1) + " times";
    return ALTRO_INVALID_ARG;
  }
  h->spec.tk.assign(tk, tk + count);  // (only a time-varying model reads them; the steps decide which kernels run)
  if (h->uploaded) return h->engine->SetKnotTimes(h->spec, &h->err);
  return ALTRO_OK;
}
altro_status altro_get_steps(altro_handle h, float* hk, float* tk) {
  if (!h) return ALTRO_INVALID_ARG;
  const int N = h->spec.desc.N;
  for (int k = 0; k < N && hk; ++k) hk[k] = h->spec.hk.empty() ? h->spec.hstep : h->spec.hk[k];
  for (int k = 0; k <= N && tk; ++k)
    tk[k] = !h->spec.tk.empty() ? h->spec.tk[k]
                                : (!h->spec.hk.empty() ? 0.0f : (k < N ? static_cast<float>(k) * h->spec.hstep : h->spec.hstep * N));
  return ALTRO_OK;
}
altro_status altro_set_lqr_cost(altro_handle h, int k_begin, int k_end, const double* Q, const double* R,
                                const double* xref, const double* uref, int per_instance) {
  if (!h || !Q || !R || !xref || !uref) return ALTRO_INVALID_ARG;
  if (DefChanged(h) != ALTRO_OK) return ALTRO_NOT_READY;
  const altro_desc& d = h->spec.desc;
  if (k_begin < 0 || k_end > d.N + 1 || k_begin >= k_end) {
    h->err = "knot range out of bounds";
    return ALTRO_INVALID_ARG;
  }
  CostSpec c;
  c.k_begin = k_begin;
  c.k_end = k_end;
  c.per_instance = per_instance;
  c.Q.assign(Q, Q + d.n * d.n);
  c.R.assign(R, R + d.m * d.m);
  c.xref.assign(xref, xref + (size_t)d.n * ((per_instance & 1) ? d.batch : 1));
  c.uref.assign(uref, uref + (size_t)d.m * ((per_instance & 2) ? d.batch : 1));
  h->spec.costs.push_back(std::move(c));
  return ALTRO_OK;
}
altro_status altro_set_user_cost(altro_handle h, int k_begin, int k_end, const double* params, int nparams,
                                 int per_instance) {
  return altro_set_user_cost_type(h, 0, k_begin, k_end, params, nparams, per_instance);
}
altro_status altro_set_user_cost_type(altro_handle h, int type, int k_begin, int k_end, const double* params, int nparams,
                                      int per_instance) {
  if (!h || type < 0 || nparams < 0 || (nparams > 0 && !params)) return ALTRO_INVALID_ARG;
  if (DefChanged(h) != ALTRO_OK) return ALTRO_NOT_READY;
  const altro_desc& d = h->spec.desc;
  if (k_begin < 0 || k_end > d.N + 1 || k_begin >= k_end) {
    h->err = "knot range out of bounds";
    return ALTRO_INVALID_ARG;
  }
  CostSpec c;
  c.k_begin = k_begin;
  c.k_end = k_end;
  c.per_instance = per_instance ? 1 : 0;
  c.user = 1 + type;
  if (nparams > 0) c.params.assign(params, params + (size_t)nparams * (per_instance ? d.batch : 1));
  h->spec.costs.push_back(std::move(c));  // the type's nparams is checked when the engine of the model exists
  return ALTRO_OK;
}
altro_status altro_add_user_constraint_type(altro_handle h, int type, int k_begin, int k_end, const double* params, int nparams,
                                            int per_instance) {
  if (!h || type < 0) return ALTRO_INVALID_ARG;
  const altro_status st = altro_add_constraint(h, ALTRO_CON_USER, k_begin, k_end, params, nparams, per_instance);
  if (st == ALTRO_OK) h->spec.cons.back().user_type = type;
  return st;
}
altro_status altro_add_constraint(altro_handle h, int kind, int k_begin, int k_end, const double* params,
                                  int nparams, int per_instance) {
  if (!h || nparams < 0 || (nparams > 0 && !params) || (nparams == 0 && kind != ALTRO_CON_USER)) return ALTRO_INVALID_ARG;
  if (DefChanged(h) != ALTRO_OK) return ALTRO_NOT_READY;
  const altro_desc& d = h->spec.desc;
  if (k_begin < 0 || k_end > d.N + 1 || k_begin >= k_end) {
    h->err = "knot range out of bounds";
    return ALTRO_INVALID_ARG;
  }
  if (kind == ALTRO_CON_CONTROL_BOUND) {
    // ControlBound::ValidateBounds, examples/basic_constraints.hpp:131-136
    if (nparams != 2 * d.m) {
      h->err = "control bound needs lb[m], ub[m]";
      return ALTRO_INVALID_ARG;
    }
    for (int j = 0; j < d.m; ++j)
      if (!(params[j] <= params[d.m + j])) {
        h->err = "Lower bound isn't less than the upper bound.";
        return ALTRO_INVALID_ARG;
      }
  }
  ConSpec c;
  c.kind = kind;
  c.k_begin = k_begin;
  c.k_end = k_end;
  c.nparams = nparams;
  c.per_instance = per_instance;
  if (nparams > 0) c.params.assign(params, params + (size_t)nparams * (per_instance ? d.batch : 1));
  h->spec.cons.push_back(std::move(c));
  return ALTRO_OK;
}
altro_status altro_set_initial_state(altro_handle h, const double* x0, int per_instance) {
  if (!h || !x0) return ALTRO_INVALID_ARG;
  if (Busy(h)) return ALTRO_NOT_READY;
  const altro_desc& d = h->spec.desc;
  h->spec.x0.assign(x0, x0 + (size_t)d.n * (per_instance ? d.batch : 1));
  h->spec.x0_per_instance = per_instance;
  if (h->uploaded) return h->engine->SetInitialState(h->spec, &h->err);
  return ALTRO_OK;
}
altro_status altro_set_trajectory(altro_handle h, const double* X, const double* U, int per_instance) {
  if (!h) return ALTRO_INVALID_ARG;
  if (Busy(h)) return ALTRO_NOT_READY;
  const altro_desc& d = h->spec.desc;
  const size_t mult = per_instance ? d.batch : 1;
  h->spec.has_X = X != nullptr;
  h->spec.has_U = U != nullptr;
  h->spec.traj_per_instance = per_instance;
  if (h->uploaded) {  // (no copy into the recorded definition: nothing reads it after the upload)
    h->spec.X_view = X;
    h->spec.U_view = U;
    const altro_status st = h->engine->SetTrajectory(h->spec, &h->err);
    h->spec.X_view = nullptr;
    h->spec.U_view = nullptr;
    return st;
  }
  if (X) h->spec.X.assign(X, X + mult * (d.N + 1) * d.n);
  if (U) h->spec.U.assign(U, U + mult * d.N * d.m);
  return ALTRO_OK;
}
altro_status altro_reset_trajectory(altro_handle h) {
  return Forward(h, [&](EngineBase& e) { return e.ResetTrajectory(); });
}
altro_status altro_reset_stats(altro_handle h) {
  return Forward(h, [&](EngineBase& e) { return e.ResetStats(); });
}
altro_status altro_set_options(altro_handle h, const altro_options* o) {
  if (!h || !o) return ALTRO_INVALID_ARG;
  if (Busy(h)) return ALTRO_NOT_READY;
  h->opts = *o;
  return ALTRO_OK;
}
altro_status altro_get_options(altro_handle h, altro_options* o) {
  if (!h || !o) return ALTRO_INVALID_ARG;
  *o = h->opts;
  return ALTRO_OK;
}
altro_status altro_set_penalty(altro_handle h, double rho) {
  if (!h || !(rho >= 0)) return ALTRO_INVALID_ARG;  // ALTRO_ASSERT(rho >= 0), constraint_values.hpp:80
  if (Busy(h)) return ALTRO_NOT_READY;
  h->spec.penalty = rho;
  if (h->uploaded) return Forward(h, [&](EngineBase& e) { return e.SetPenalty(rho); });
  return ALTRO_OK;
}
altro_status altro_set_penalty_scaling(altro_handle h, double phi) {
  if (!h || !(phi >= 1)) return ALTRO_INVALID_ARG;  // ALTRO_ASSERT(phi >= 1), constraint_values.hpp:85
  if (Busy(h)) return ALTRO_NOT_READY;
  h->spec.phi = phi;
  if (h->uploaded) return Forward(h, [&](EngineBase& e) { return e.SetPenaltyScaling(phi); });
  return ALTRO_OK;
}

altro_status altro_solve_al(altro_handle h) {
  if (h) h->ilqr_mode = false;
  return Forward(h, [&](EngineBase& e) { return e.SolveAL(h->opts); });
}
altro_status altro_solve_ilqr(altro_handle h) {
  if (h) h->ilqr_mode = true;
  return Forward(h, [&](EngineBase& e) { return e.SolveILQR(h->opts); });
}
altro_status altro_solve_al_async(altro_handle h) {
  if (!h) return ALTRO_INVALID_ARG;
  if (Busy(h)) return ALTRO_NOT_READY;
  // create the device state on the caller's thread: definition errors are reported here, synchronously
  altro_status st = Ensure(h);
  if (st != ALTRO_OK) return st;
  h->ilqr_mode = false;
  if (!h->worker.joinable()) {
    h->worker = std::thread([h]() {
      for (;;) {
        {
          std::unique_lock<std::mutex> lk(h->mu);
          h->cv.wait(lk, [h]() { return h->job_posted || h->worker_exit; });
          if (h->worker_exit) return;
          h->job_posted = false;
        }
        // the engine and the options are frozen while async_pending is set (Busy() guards every setter)
        h->async_status = h->engine->SolveAL(h->opts);
        if (h->async_status != ALTRO_OK) h->async_err = h->engine->LastError();
        {
          std::lock_guard<std::mutex> lk(h->mu);  // (the flag changes under the mutex: altro_wait cannot miss the wake-up)
          h->async_done.store(1, std::memory_order_release);
        }
        h->cv_done.notify_all();
      }
    });
  }
  h->async_pending = true;
  h->async_done.store(0);
  {
    std::lock_guard<std::mutex> lk(h->mu);
    h->job_posted = true;
  }
  h->cv.notify_one();
  return ALTRO_OK;
}
altro_status altro_solve_poll(altro_handle h, int* done) {
  if (!h || !done) return ALTRO_INVALID_ARG;
  *done = h->async_done.load(std::memory_order_acquire);
  return ALTRO_OK;
}
altro_status altro_wait(altro_handle h) {
  if (!h) return ALTRO_INVALID_ARG;
  if (!h->async_pending) {
    h->err = "no asynchronous solve is pending";
    return ALTRO_NOT_READY;
  }
  {
    // blocks on a condition variable (the atomic alone serves altro_solve_poll): a waiting caller costs no core
    std::unique_lock<std::mutex> lk(h->mu);
    h->cv_done.wait(lk, [h]() { return h->async_done.load(std::memory_order_acquire) != 0; });
  }
  h->async_pending = false;
  if (h->async_status != ALTRO_OK) h->err = h->async_err;
  return h->async_status;
}
altro_status altro_al_init(altro_handle h) { return Forward(h, [&](EngineBase& e) { return e.AlInit(h->opts); }); }
altro_status altro_solve_setup(altro_handle h) { return Forward(h, [&](EngineBase& e) { return e.SolveSetup(h->opts); }); }
altro_status altro_rollout(altro_handle h) { return Forward(h, [&](EngineBase& e) { return e.Rollout(h->opts); }); }
altro_status altro_cost(altro_handle h, double* J) { return Forward(h, [&](EngineBase& e) { return e.Cost(h->opts, J); }); }
altro_status altro_update_expansions(altro_handle h) { return Forward(h, [&](EngineBase& e) { return e.UpdateExpansions(h->opts); }); }
altro_status altro_backward_pass(altro_handle h) { return Forward(h, [&](EngineBase& e) { return e.BackwardPass(h->opts); }); }
altro_status altro_forward_pass(altro_handle h) { return Forward(h, [&](EngineBase& e) { return e.ForwardPass(h->opts); }); }
altro_status altro_update_convergence_statistics(altro_handle h) {
  return Forward(h, [&](EngineBase& e) { return e.UpdateConvergenceStatistics(h->opts); });
}
altro_status altro_update_duals(altro_handle h) { return Forward(h, [&](EngineBase& e) { return e.UpdateDuals(h->opts); }); }
altro_status altro_update_penalties(altro_handle h) { return Forward(h, [&](EngineBase& e) { return e.UpdatePenalties(h->opts); }); }
altro_status altro_get_max_violation(altro_handle h, double* out) {
  if (!out) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.GetMaxViolation(out); });
}
altro_status altro_max_violation(altro_handle h, double* out) {
  if (!out) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) {
    altro_status st = e.Cost(h->opts, nullptr);
    return st != ALTRO_OK ? st : e.GetMaxViolation(out);
  });
}
altro_status altro_get_max_penalty(altro_handle h, double* out) {
  if (!out) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.GetMaxPenalty(out); });
}

altro_status altro_get_trajectory(altro_handle h, double* X, double* U) {
  return Forward(h, [&](EngineBase& e) { return e.GetTrajectory(X, U); });
}
altro_status altro_get_gains(altro_handle h, double* K, double* d) {
  return Forward(h, [&](EngineBase& e) { return e.GetGains(K, d); });
}
altro_status altro_set_record_ctg(altro_handle h, int enable) {
  return Forward(h, [&](EngineBase& e) { return e.SetRecordCtg(enable); });
}
altro_status altro_get_ctg(altro_handle h, double* P, double* p) {
  return Forward(h, [&](EngineBase& e) { return e.GetCtg(P, p); });
}
altro_status altro_get_expansion(altro_handle h, int k, double* AB, double* lxx, double* lxu, double* luu,
                                 double* lx, double* lu) {
  return Forward(h, [&](EngineBase& e) { return e.GetExpansion(k, AB, lxx, lxu, luu, lx, lu); });
}
altro_status altro_get_knot_costs(altro_handle h, double* costs) {
  if (!costs) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.GetKnotCosts(costs); });
}
int altro_num_constraints(altro_handle h) {
  if (!h || Busy(h) || Ensure(h) != ALTRO_OK) return -1;
  return h->engine->NumRows();
}
int altro_num_constraints_at(altro_handle h, int k) {
  if (!h || Busy(h) || Ensure(h) != ALTRO_OK) return -1;
  return h->engine->NumRowsAt(k);
}
altro_status altro_get_duals(altro_handle h, double* lam) {
  if (!lam) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.GetRows(0, lam); });
}
altro_status altro_set_duals(altro_handle h, const double* lam) {
  if (!lam) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.SetDuals(lam); });
}
altro_status altro_get_penalties(altro_handle h, double* rho) {
  if (!rho) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.GetRows(1, rho); });
}
altro_status altro_get_constraint_values(altro_handle h, double* c) {
  if (!c) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.GetRows(2, c); });
}
altro_status altro_get_stats(altro_handle h, altro_stats* stats) {
  if (!stats) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.GetStats(stats, h->ilqr_mode); });
}
altro_status altro_get_timing(altro_handle h, altro_timing* t) {
  if (!t) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.GetTiming(t); });
}
altro_status altro_set_record_history(altro_handle h, int capacity) {
  return Forward(h, [&](EngineBase& e) { return e.SetRecordHistory(capacity); });
}
int altro_get_history(altro_handle h, int instance, int field, double* out, int cap) {
  if (!h || !out || Busy(h) || Ensure(h) != ALTRO_OK) return -1;
  return h->engine->GetHistory(instance, field, out, cap);
}
int altro_get_history_all(altro_handle h, int instance, double* out, int cap) {
  if (!h || !out || Busy(h) || Ensure(h) != ALTRO_OK) return -1;
  return h->engine->GetHistoryAll(instance, out, cap);
}
altro_status altro_device_info(altro_handle h, char* name, int name_len, int* cu_count) {
  if (!h) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.DeviceInfo(name, name_len, cu_count); });
}
altro_status altro_pack_results_device(altro_handle h, void* dst_device) {
  if (!dst_device) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.PackResultsDevice(dst_device); });
}
altro_status altro_pack_trajectory_device(altro_handle h, void* X_device, void* U_device) {
  if (!X_device && !U_device) return ALTRO_INVALID_ARG;
  return Forward(h, [&](EngineBase& e) { return e.PackTrajectoryDevice((double*)X_device, (double*)U_device); });
}

}  // extern "C"
