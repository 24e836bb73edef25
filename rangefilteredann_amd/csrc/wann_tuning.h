// wann_tuning.h -- every development / test switch of the engine as ONE typed record.
//
// The record is filled from the environment ONCE, when an index (or a raw graph) is created; the batch_search call never
// reads the environment.  None of the switches changes a result row (the parity tests run them against each other).
//   * A PRODUCTION process honours three things: WANN_VERBOSE (launch lines on stderr), WANN_PROOF_FACTOR (can only widen the
//     dense path's proof margin) and the HIP runtime's own serialisation variables.  (WANN_DEVICES -- which GPUs an index is
//     replicated on -- and WANN_ASYNC_LANES -- batches in flight of the asynchronous call -- are read by the C ABI, not here.)
//   * The LABORATORY (fifteen names, lab_names() below) only counts with WANN_TEST_HOOKS=1 (a development switch found in the
//     environment without it is named once on stderr; tests/conftest.py and the tools set it): switches that turn a mechanism OFF
//     so that tests can compare rows with and without it (speculative levels, companion launch, pollers, look-aheads, helper
//     waves, the dense path, sliced scans), hooks that FORCE rare paths (pollers under serialised dispatch, pollers that give up,
//     eager look-aheads, the general cores at small beams, the first-generation core, raw searches in the one-wave kernel) and
//     one threshold (what counts as a saturated launch).  Round 6 removed twenty scheduling knobs whose A/B runs are on record in
//     profiles/ (yield, gate, evidence-first, order, lean pool, scan thresholds, poller counts, priorities, caps ...): what they
//     selected is now a constant next to the code that uses it.
//   * With WANN_TEST_HOOKS=1 the entry points of the C ABI also re-read the record before every call (tests flip switches
//     between batches on one index); without it the record is fixed for the life of the index.
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

extern "C" char **environ;  // (POSIX; unistd.h declares it only under _GNU_SOURCE)

namespace wann {

struct Tuning {
  bool hooks_live = false;  // WANN_TEST_HOOKS=1: re-read before every call, the laboratory honoured
  // mechanisms a test can switch off
  bool spec = true;            // !WANN_NO_SPEC: speculative doubling levels
  bool big = true;             // !WANN_NO_BIG: companion launch of the one-wave kernel
  bool pollers = true;         // !WANN_NO_POLLERS
  bool helper = true;          // !WANN_NO_HELPER: scoring helper waves of the one-wave kernel
  bool lookahead = true;       // !WANN_NO_LOOKAHEAD (also the idle pollers' scan)
  bool split_scan = true;      // !WANN_NO_SPLIT_SCAN
  bool gemm = true;            // !WANN_NO_GEMM
  bool dense_always = false;   // WANN_DENSE_ALWAYS
  long long deep_min_tasks = 4096;  // WANN_DEEP_MIN_TASKS: graph tasks from which a launch counts as saturated (deep-chain pollers)
  bool serialized = false;     // HIP_LAUNCH_BLOCKING / AMD_SERIALIZE_KERNEL / CUDA_LAUNCH_BLOCKING: launches never overlap
  float proof_factor = 3.f;    // WANN_PROOF_FACTOR, clamped to >= 3 (see dense_prefilter)
  // hooks that force rare paths
  bool force_pollers = false, force_poll_timeout = false, la_eager = false, force_general = false, old_general = false,
       raw_big_lds = false;
  // diagnostics
  bool verbose = false;         // WANN_VERBOSE
  bool profile_phases = false;  // WANN_PROFILE_PHASES (make PROFILE=1 builds)
  std::string task_trace;       // WANN_TASK_TRACE=<file> (make TRACE=1 builds)

  static bool on(const char *name) {
    const char *v = getenv(name);
    return v && *v && strcmp(v, "0") != 0;
  }
  static bool set(const char *name) { return getenv(name) != nullptr; }
  // the laboratory: names that only count with WANN_TEST_HOOKS=1 (+ the two diagnostics of the dev builds)
  static const char *const *lab_names() {
    static const char *const names[] = {
        "WANN_NO_SPEC", "WANN_NO_BIG", "WANN_NO_POLLERS", "WANN_NO_HELPER", "WANN_NO_LOOKAHEAD", "WANN_NO_SPLIT_SCAN", "WANN_NO_GEMM", "WANN_DENSE_ALWAYS",
        "WANN_DEEP_MIN_TASKS", "WANN_FORCE_POLLERS", "WANN_FORCE_POLL_TIMEOUT", "WANN_LA_EAGER", "WANN_FORCE_GENERAL", "WANN_OLD_GENERAL", "WANN_RAW_BIG_LDS",
        "WANN_PROFILE_PHASES", "WANN_TASK_TRACE", nullptr};
    return names;
  }
  // every WANN_* name this library or its Python package reads (beside the laboratory): anything else in the environment is a
  // typo or a switch of another version -- named once on stderr (a sweep over a variable nothing reads ran silently)
  static bool known_name(const char *name, size_t len) {
    static const char *const names[] = {"WANN_TEST_HOOKS", "WANN_VERBOSE", "WANN_PROOF_FACTOR", "WANN_DEVICES", "WANN_DEVICE", "WANN_ASYNC_LANES",
                                        "WANN_HOST_BUILD", "WANN_REF_TIES", "WANN_BUILD_VIS_CAP", "WANN_NO_TORCH", "WANN_DATASET_FOLDER", "WANN_LIB", nullptr};
    static const char *const prefixes[] = {"WANN_BENCH_", "WANN_PF_", "WANN_FULLSIZE_", nullptr};  // (bench.py / tools / tests)
    for (const char *const *n = names; *n; n++)
      if (strlen(*n) == len && !strncmp(*n, name, len)) return true;
    for (const char *const *n = lab_names(); *n; n++)
      if (strlen(*n) == len && !strncmp(*n, name, len)) return true;
    for (const char *const *n = prefixes; *n; n++)
      if (len >= strlen(*n) && !strncmp(*n, name, strlen(*n))) return true;
    return false;
  }
  static void warn_unknown_names() {
    static std::atomic<bool> done{false};
    if (done.exchange(true)) return;
    std::string found;
    for (char **e = ::environ; e && *e; e++) {
      if (strncmp(*e, "WANN_", 5) != 0) continue;
      const char *eq = strchr(*e, '=');
      const size_t len = eq ? (size_t)(eq - *e) : strlen(*e);
      if (!known_name(*e, len)) found += std::string(found.empty() ? "" : ", ") + std::string(*e, len);
    }
    if (!found.empty()) fprintf(stderr, "[wann] unknown WANN_* variable(s) in the environment (nothing reads them): %s\n", found.c_str());
  }
  static Tuning from_env() {
    Tuning t;
    warn_unknown_names();
    t.hooks_live = on("WANN_TEST_HOOKS");
    t.serialized = on("HIP_LAUNCH_BLOCKING") || on("AMD_SERIALIZE_KERNEL") || on("CUDA_LAUNCH_BLOCKING");
    t.verbose = set("WANN_VERBOSE");
    // The proof's accumulation term: 3 = the truncating-adder bound (2 d u |q||p| per product term) and half as much again.
    // A smaller factor would let k_rerank certify results it has not proven, so the knob can only WIDEN the margin.
    if (const char *v = getenv("WANN_PROOF_FACTOR")) {
      const float f = (float)atof(v);
      t.proof_factor = (f == f && f > 3.f) ? f : 3.f;
    }
    if (!t.hooks_live) {
      static std::atomic<bool> told{false};
      if (!told.load(std::memory_order_relaxed)) {
        std::string found;
        for (const char *const *n = lab_names(); *n; n++)
          if (getenv(*n)) found += std::string(found.empty() ? "" : ", ") + *n;
        if (!found.empty() && !told.exchange(true))
          fprintf(stderr, "[wann] development switch(es) in the environment are IGNORED without WANN_TEST_HOOKS=1: %s\n", found.c_str());
      }
      return t;
    }
    t.spec = !set("WANN_NO_SPEC");
    t.big = !set("WANN_NO_BIG");
    t.pollers = !set("WANN_NO_POLLERS");
    t.helper = !set("WANN_NO_HELPER");
    t.lookahead = !set("WANN_NO_LOOKAHEAD");
    t.split_scan = !set("WANN_NO_SPLIT_SCAN");
    t.gemm = !set("WANN_NO_GEMM");
    t.dense_always = set("WANN_DENSE_ALWAYS");
    if (const char *v = getenv("WANN_DEEP_MIN_TASKS")) t.deep_min_tasks = atoll(v);
    t.force_pollers = set("WANN_FORCE_POLLERS");
    t.force_poll_timeout = set("WANN_FORCE_POLL_TIMEOUT");
    t.la_eager = set("WANN_LA_EAGER");
    t.force_general = set("WANN_FORCE_GENERAL");
    t.old_general = set("WANN_OLD_GENERAL");
    t.raw_big_lds = set("WANN_RAW_BIG_LDS");
    t.profile_phases = set("WANN_PROFILE_PHASES");
    if (const char *v = getenv("WANN_TASK_TRACE")) t.task_trace = v;
    return t;
  }
};

}  // namespace wann
