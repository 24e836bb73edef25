// wann_tuning.h -- every development / test switch of the engine as ONE typed record.
//
// The record is filled from the environment ONCE, when an index (or a raw graph) is created; the batch_search call never
// reads the environment.  None of the switches changes a result row (the parity tests run them against each other).
//   * A PRODUCTION process honours three things: WANN_VERBOSE (launch lines on stderr), WANN_PROOF_FACTOR (can only widen the
//     dense path's proof margin) and the HIP runtime's own serialisation variables.  (WANN_DEVICES -- which GPUs an index is
//     replicated on -- is read by the C ABI, not here.)
//   * EVERYTHING ELSE -- the scheduling / launch-shape knobs (WANN_NO_SPEC, WANN_POLLERS, WANN_HEAVY_RATIO ...) and the hooks that
//     force rare paths in tests (WANN_FORCE_POLLERS, WANN_FORCE_POLL_TIMEOUT, WANN_LA_EAGER, WANN_FORCE_GENERAL,
//     WANN_OLD_GENERAL, WANN_RAW_BIG_LDS, WANN_BUILD_VIS_CAP) -- is the laboratory: ignored unless WANN_TEST_HOOKS=1 (round 5; a
//     development switch found in the environment without it is named once on stderr).  tests/conftest.py and the tools set it.
//   * With WANN_TEST_HOOKS=1 the entry points of the C ABI also re-read the record before every call (tests flip switches
//     between batches on one index); without it the record is fixed for the life of the index.
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace wann {

struct Tuning {
  bool hooks_live = false;  // WANN_TEST_HOOKS=1: re-read before every call, test-only hooks honoured
  // k_route / scheduling
  bool spec = true;            // !WANN_NO_SPEC: speculative doubling levels
  bool big = true;             // !WANN_NO_BIG: companion launch of the one-wave kernel
  bool pollers = true;         // !WANN_NO_POLLERS
  bool serialized = false;     // HIP_LAUNCH_BLOCKING / AMD_SERIALIZE_KERNEL / CUDA_LAUNCH_BLOCKING: launches never overlap
  bool yield = true;           // !WANN_NO_YIELD
  bool helper = true;          // !WANN_NO_HELPER: scoring helper waves of the one-wave kernel
  bool deep = true;            // !WANN_NO_DEEP
  bool gate = true;            // !WANN_NO_GATE: the ordinary launch waits for the deep-chain pollers to start
  bool lookahead = true;       // !WANN_NO_LOOKAHEAD
  bool scan = true;            // WANN_SCAN != 0
  bool evidence_first = true;  // !WANN_NO_EVIDENCE_FIRST
  bool order = true;           // !WANN_NO_ORDER
  bool lean = true;            // !WANN_NO_LEAN
  bool split_scan = true;      // !WANN_NO_SPLIT_SCAN
  bool gemm = true;            // !WANN_NO_GEMM
  bool dense_always = false;   // WANN_DENSE_ALWAYS
  int heavy_ratio = 8;         // WANN_HEAVY_RATIO
  int spec_num = 8;            // WANN_SPEC_NUM
  int spec_extra = 2;          // WANN_SPEC_EXTRA (see RouteArgs::spec_extra; 0: off)
  int npollers = 0;            // WANN_POLLERS (0: 32 with the scan, else 16)
  int deep_pollers = 0;        // WANN_DEEP_POLLERS (0: 4, or 16 where three workgroups share a CU)
  long long deep_min_tasks = 4096;  // WANN_DEEP_MIN_TASKS
  int scan_num = 16;           // WANN_SCAN_NUM
  int scan_min_top = 2560;     // WANN_SCAN_MIN_TOP
  int big_exclusive = -1;      // WANN_BIG_EXCLUSIVE (-1: by launch kind)
  int blocks_per_cu = 0;       // WANN_BLOCKS_PER_CU (0: by registers / LDS)
  int lean_pool = 0;           // WANN_LEAN_POOL (0: computed)
  int brute_per_cu = 0;        // WANN_BRUTE_PER_CU (0: by element type)
  int search_prio = 0;         // WANN_SEARCH_PRIO
  int inkernel_cap = 0;        // WANN_INKERNEL_CAP (0: kInKernelBeamCap): largest beam of the four-wave kernel; levels above go to the companion launch
  float proof_factor = 3.f;    // WANN_PROOF_FACTOR, clamped to >= 3 (see dense_prefilter)
  // test-only hooks (WANN_TEST_HOOKS=1)
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
  static int num(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
  }
  // the laboratory: names that only count with WANN_TEST_HOOKS=1
  static const char *const *lab_names() {
    static const char *const names[] = {
        "WANN_NO_SPEC", "WANN_NO_BIG", "WANN_NO_POLLERS", "WANN_NO_YIELD", "WANN_NO_HELPER", "WANN_NO_DEEP", "WANN_NO_GATE", "WANN_NO_LOOKAHEAD",
        "WANN_SCAN", "WANN_NO_EVIDENCE_FIRST", "WANN_NO_ORDER", "WANN_NO_LEAN", "WANN_NO_SPLIT_SCAN", "WANN_NO_GEMM", "WANN_DENSE_ALWAYS",
        "WANN_HEAVY_RATIO", "WANN_SPEC_NUM", "WANN_SPEC_EXTRA", "WANN_POLLERS", "WANN_DEEP_POLLERS", "WANN_DEEP_MIN_TASKS", "WANN_SCAN_NUM", "WANN_SCAN_MIN_TOP",
        "WANN_BIG_EXCLUSIVE", "WANN_INKERNEL_CAP", "WANN_BLOCKS_PER_CU", "WANN_LEAN_POOL", "WANN_BRUTE_PER_CU", "WANN_SEARCH_PRIO", "WANN_FORCE_POLLERS",
        "WANN_FORCE_POLL_TIMEOUT", "WANN_LA_EAGER", "WANN_FORCE_GENERAL", "WANN_OLD_GENERAL", "WANN_RAW_BIG_LDS", "WANN_PROFILE_PHASES",
        "WANN_TASK_TRACE", nullptr};
    return names;
  }
  static Tuning from_env() {
    Tuning t;
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
    t.yield = !set("WANN_NO_YIELD");
    t.helper = !set("WANN_NO_HELPER");
    t.deep = !set("WANN_NO_DEEP");
    t.gate = !set("WANN_NO_GATE");
    t.lookahead = !set("WANN_NO_LOOKAHEAD");
    t.scan = !(set("WANN_SCAN") && num("WANN_SCAN", 1) == 0);
    t.evidence_first = !set("WANN_NO_EVIDENCE_FIRST");
    t.order = !set("WANN_NO_ORDER");
    t.lean = !set("WANN_NO_LEAN");
    t.split_scan = !set("WANN_NO_SPLIT_SCAN");
    t.gemm = !set("WANN_NO_GEMM");
    t.dense_always = set("WANN_DENSE_ALWAYS");
    t.heavy_ratio = num("WANN_HEAVY_RATIO", 8);
    if (t.heavy_ratio < 1) t.heavy_ratio = 1;
    t.spec_num = num("WANN_SPEC_NUM", 8);
    if (t.spec_num < 1) t.spec_num = 1;
    t.spec_extra = num("WANN_SPEC_EXTRA", 2);
    t.npollers = num("WANN_POLLERS", 0);
    if (t.npollers < 0) t.npollers = 0;
    t.deep_pollers = num("WANN_DEEP_POLLERS", 0);
    if (t.deep_pollers < 0) t.deep_pollers = 0;
    if (const char *v = getenv("WANN_DEEP_MIN_TASKS")) t.deep_min_tasks = atoll(v);
    t.scan_num = num("WANN_SCAN_NUM", 16);
    t.scan_min_top = num("WANN_SCAN_MIN_TOP", 2560);
    t.big_exclusive = set("WANN_BIG_EXCLUSIVE") ? (num("WANN_BIG_EXCLUSIVE", 0) != 0 ? 1 : 0) : -1;
    t.blocks_per_cu = num("WANN_BLOCKS_PER_CU", 0);
    t.lean_pool = num("WANN_LEAN_POOL", 0);
    t.brute_per_cu = num("WANN_BRUTE_PER_CU", 0);
    t.search_prio = num("WANN_SEARCH_PRIO", 0);
    t.inkernel_cap = num("WANN_INKERNEL_CAP", 0);
    if (t.inkernel_cap < 0 || t.inkernel_cap > 1280) t.inkernel_cap = 0;  // (the four-wave kernel's LDS pool holds beams up to 1 280)
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
