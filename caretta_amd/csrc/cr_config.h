// Calibration switches of libcaretta_hip, read from the environment ONCE (when the library is loaded) into one struct.
//
// None of these is part of the product's interface: they exist so that the measurement tools (tools/c3_share.py,
// tools/calibrate_*.py, tools/stamps.py) and the parity tests can force a kernel family, a strip plan or a limit and
// compare it with the library's own choice on the same box.  A process that changes its environment afterwards (those
// tools do, between two pair lists) calls cr_config_reload() -- engine.reload_config() -- to have the change seen;
// nothing on a hot path ever calls getenv.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace crcfg {

inline const char* env_str(const char* name) { return std::getenv(name); }
inline bool env_set(const char* name) { return std::getenv(name) != nullptr; }
// "0" switches a default-on feature off (CARETTA_STAGED=0, CARETTA_TRIO=0, CARETTA_MID=0)
inline bool env_on(const char* name) {
    const char* e = std::getenv(name);
    return !(e && e[0] == '0');
}
inline long long env_ll(const char* name, long long dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoll(e) : dflt;
}

struct StripPlanEnv {
    bool set = false;
    int ra = 0, rb = 0, na = 0, sync = 0;
};

struct Calibration {
    // ---- which kernel family serves a pair list (cr_batch_set_pairs: choose_layout) ----
    bool no_team = false;            // CARETTA_NO_TEAM: no multi-wave layouts at all
    bool no_wide = false;            // CARETTA_NO_WIDE
    bool sw_rows_nowalk = false;     // CARETTA_SW_ROWS_NOWALK: measurement only -- the fill of k_sw_trace_rows without its walk (results are then wrong)
    int sw_rows_waves = 0;           // CARETTA_SW_ROWS_WAVES=3|5: measurement only -- k_sw_trace_rows<5> built for that many waves per SIMD
    bool no_sw_rows = false;         // CARETTA_NO_SW_ROWS: smith_waterman lists with gap 0 on the skewed sweep + walk launch (the path before round 6)
    bool trio = true;                // CARETTA_TRIO=0 switches the split by function off
    bool mid = true;                 // CARETTA_MID=0 switches the mid-size row split off
    bool mid_any = false;            // CARETTA_MID_ANY: k_pair_duo also with a single strip (measurements)
    bool staged = true;              // CARETTA_STAGED=0: the fused kernels instead of staged scores
    bool classes = true;             // CARETTA_CLASSES=0: a ragged list stays ONE list (no size classes)
    long long team_pairs = -1;       // CARETTA_TEAM_PAIRS   (-1: the table's limit)
    long long trio_pairs = -1;       // CARETTA_TRIO_PAIRS
    long long trio_from = -1;        // CARETTA_TRIO_FROM
    long long trio_min_rows = -1;    // CARETTA_TRIO_MIN_ROWS
    long long mid_pairs = -1;        // CARETTA_MID_PAIRS
    long long staged_waves = -1;     // CARETTA_STAGED_WAVES
    long long staged_rows = -1;      // CARETTA_STAGED_ROWS
    StripPlanEnv wide;               // CARETTA_WIDE=RA,RB,nA,B
    StripPlanEnv mid_plan;           // CARETTA_MID_PLAN=RA,RB,nA
    // ---- launch shapes ----
    int trio_waves = 0;              // CARETTA_TRIO_WAVES  (0: the library's choice)
    int trio_waves2 = 0;             // CARETTA_TRIO_WAVES2 (second stage)
    int mid_lds_kb = 0;              // CARETTA_MID_LDS_KB: pad the dynamic LDS (pairs per CU)
    int force_r = 0;                 // CARETTA_FORCE_R: rows per lane of the single-wave kernels
    bool keep_order = false;         // CARETTA_KEEP_ORDER: one group, the caller's order
    long long scratch_mb = 0;        // CARETTA_SCRATCH_MB: decision scratch per chunk
    // ---- explicit-matrix batches ----
    int stream_lds_kb = 0;           // CARETTA_STREAM_LDS_KB
    int stream_r = 0;                // CARETTA_STREAM_R
    // ---- single calls, neighbor joining, progressive alignment, multi-GPU ----
    long long host_small_k = 4096;   // CARETTA_HOST_SMALL_K
    int nj_threads = 0;              // CARETTA_NJ_THREADS
    int nj_groups = 0;               // CARETTA_NJ_GROUPS
    bool nj_profile = false;         // CARETTA_NJ_PROFILE
    bool nj_device_strict = false;   // CARETTA_NJ_DEVICE_STRICT
    bool sync_levels = false;        // CARETTA_SYNC_LEVELS
    bool multi_allow_duplicates = false;   // CARETTA_MULTI_ALLOW_DUPLICATES
    bool multi_numa = false;         // CARETTA_MULTI_NUMA=1: pin each device's host thread to the CPUs of the device's NUMA node
    std::string rccl_lib;            // CARETTA_RCCL_LIB
    long long cache_mb = 0;          // CARETTA_CACHE_MB
    bool no_cache = false;           // CARETTA_NO_CACHE

    static StripPlanEnv plan_from(const char* name, int fields) {
        StripPlanEnv p;
        if (const char* e = std::getenv(name)) {
            int got = 0;
            if (fields == 4) got = std::sscanf(e, "%d,%d,%d,%d", &p.ra, &p.rb, &p.na, &p.sync);
            else got = std::sscanf(e, "%d,%d,%d", &p.ra, &p.rb, &p.na);
            p.set = got == fields;
        }
        return p;
    }

    static Calibration from_env() {
        Calibration c;
        c.no_team = env_set("CARETTA_NO_TEAM");
        c.no_wide = env_set("CARETTA_NO_WIDE");
        c.no_sw_rows = env_set("CARETTA_NO_SW_ROWS");
        c.sw_rows_nowalk = env_set("CARETTA_SW_ROWS_NOWALK");
        c.trio = env_on("CARETTA_TRIO");
        c.mid = env_on("CARETTA_MID");
        c.mid_any = env_set("CARETTA_MID_ANY");
        c.staged = env_on("CARETTA_STAGED");
        c.classes = env_on("CARETTA_CLASSES");
        c.team_pairs = env_ll("CARETTA_TEAM_PAIRS", -1);
        c.trio_pairs = env_ll("CARETTA_TRIO_PAIRS", -1);
        c.trio_from = env_ll("CARETTA_TRIO_FROM", -1);
        c.trio_min_rows = env_ll("CARETTA_TRIO_MIN_ROWS", -1);
        c.mid_pairs = env_ll("CARETTA_MID_PAIRS", -1);
        c.staged_waves = env_ll("CARETTA_STAGED_WAVES", -1);
        c.staged_rows = env_ll("CARETTA_STAGED_ROWS", -1);
        c.wide = plan_from("CARETTA_WIDE", 4);
        c.mid_plan = plan_from("CARETTA_MID_PLAN", 3);
        c.trio_waves = (int)env_ll("CARETTA_TRIO_WAVES", 0);
        c.trio_waves2 = (int)env_ll("CARETTA_TRIO_WAVES2", 0);
        c.mid_lds_kb = (int)env_ll("CARETTA_MID_LDS_KB", 0);
        c.force_r = (int)env_ll("CARETTA_FORCE_R", 0);
        c.keep_order = env_set("CARETTA_KEEP_ORDER");
        c.scratch_mb = env_ll("CARETTA_SCRATCH_MB", 0);
        c.stream_lds_kb = (int)env_ll("CARETTA_STREAM_LDS_KB", 0);
        c.stream_r = (int)env_ll("CARETTA_STREAM_R", 0);
        c.sw_rows_waves = (int)env_ll("CARETTA_SW_ROWS_WAVES", 0);
        c.host_small_k = env_ll("CARETTA_HOST_SMALL_K", 4096);
        c.nj_threads = (int)env_ll("CARETTA_NJ_THREADS", 0);
        c.nj_groups = (int)env_ll("CARETTA_NJ_GROUPS", 0);
        c.nj_profile = env_set("CARETTA_NJ_PROFILE");
        c.nj_device_strict = env_set("CARETTA_NJ_DEVICE_STRICT");
        c.sync_levels = env_set("CARETTA_SYNC_LEVELS");
        c.multi_allow_duplicates = env_set("CARETTA_MULTI_ALLOW_DUPLICATES");
        c.multi_numa = env_ll("CARETTA_MULTI_NUMA", 0) == 1;
        if (const char* e = env_str("CARETTA_RCCL_LIB")) c.rccl_lib = e;
        c.cache_mb = env_ll("CARETTA_CACHE_MB", 0);
        c.no_cache = env_set("CARETTA_NO_CACHE");
        return c;
    }
};

}  // namespace crcfg
