// Which kernel family serves a pair list, and whether a ragged list is split into size classes: the HOST logic of
// cr_batch_set_pairs (no device call in this file; cr_plan_layout exposes it to the CPU tests).
// Included by cr_api.hip inside its anonymous namespace, behind the launchers (StripPlan, staged_shape, padded_width, g_cfg).
//
// Reference: the reference's pair loop (multiple_alignment.py:158-170) has one code path; everything here only decides which
// kernels compute the same values.
#pragma once

// ---------------------------------------------------------------------------------------------
// Which kernel family serves a pair list: ONE table, consulted by ONE function (choose_layout).
//
// A rule applies to a list whose longest structure has rows in [rows_lo, rows_hi] and columns <= cols_hi, whose pair
// count lies in [pairs_lo, pairs_hi] and whose padded tensor width is at most d_pad_hi; the first rule that applies
// AND whose family's `fits` check passes (LDS of the launch, waves resident at once, bytes of staged scores -- what a
// range cannot say) wins.  Every limit was measured on equal-length synthetic families on an MI355X; `calibration`
// names the committed record.  The environment switches of cr_config.h move single limits for measurements.
// ---------------------------------------------------------------------------------------------
enum Family : int { kFamSingle = 0, kFamTeam, kFamWide, kFamDuo, kFamTrio, kFamStaged };

struct LayoutRule {
    Family family;
    int rows_lo, rows_hi;
    int64_t pairs_lo, pairs_hi;
    int cols_hi;
    int d_pad_hi;
    const char* calibration;
};

constexpr int64_t kAnyPairs = std::numeric_limits<int64_t>::max();
constexpr int kAnyLength = cr::kMaxLength;
constexpr int kAnyWidth = 1 << 20;
constexpr int64_t kTeamPairLimit = 256;
// Pair lists of at most this many 64-row strips run on staged scores: one wave per SIMD of the chip.
constexpr int64_t kStagedWaveLimit = 1024;
// Mid-size lists (cr_duo.h): up to this many waves (two strips per pair / more), columns resident in LDS.
constexpr int64_t kMidWaveLimit2 = 2600, kMidWaveLimit = 3072;
constexpr int64_t kTrioPairLimit = 1300;       // k_pair_trio: 1 355 pairs tie with one wave per pair
constexpr int kMidMaxColumns = 1280;            // split by function: columns resident in LDS next to the score ring (four pairs per CU)
constexpr int kDuoMaxRows = 3 * cr::kWave * (cr::kDuoMaxWaves - 1) + 2 * cr::kWave;      // 1 472: seven strips of 3 rows per lane and one of 2
constexpr int kGroupLanes = 4;                 // streams that row-per-lane groups (and size classes) are spread over
constexpr int64_t kClassSplitPairs = 4096;     // a ragged list of at most this many pairs is split into size classes

constexpr LayoutRule kLayoutTable[] = {
    // split by FUNCTION (cr_trio.h): one strip of 2 .. 5 rows per lane; its time does not depend on the pair count while the
    // chip is not full, the one-pair-per-CU layouts and staged scores grow with it -- hence "from" 65 / 161 / 161 pairs
    // (193 .. 256 rows: from 111 pairs until the staged sweeps lost the masks of their ramps and their barriers, round 5:
    // 128 pairs of 250 rows 0.258 ms on staged scores against 0.276-0.282, 160 pairs 0.280 / 0.279; profiles/r05/staged_vs_trio.txt)
    {kFamTrio, 65, 192, 65, kTrioPairLimit, kMidMaxColumns, 16, "profiles/r04/trio_few.txt, trio_sizes.txt"},
    {kFamTrio, 193, 256, 161, kTrioPairLimit, kMidMaxColumns, 16, "profiles/r05/staged_vs_trio.txt"},
    {kFamTrio, 257, 320, 161, kTrioPairLimit, kMidMaxColumns, 16, "profiles/r04/trio_few.txt, c3_share.txt, c3_share_limit.txt"},
    // staged scores (cr_staged.h): at most one wave per SIMD of the chip (pairs x strips <= 1 024: checked by fits)
    // (the only family for tensors wider than 32: its tensor scores then come from the run-time-width staging kernel)
    {kFamStaged, 1, cr::kStagedMaxRows, 1, kStagedWaveLimit, kAnyLength, kAnyWidth, "profiles/r03/calibrate_staged.txt"},
    // one pair per CU, up to 16 waves, barrier every 8 steps (k_pair_wide)
    {kFamWide, 193, 3072, 1, kTeamPairLimit, kAnyLength, 16, "profiles/r03/calibrate_wide.txt"},
    // four-wave teams: what the wide layout cannot take (tensor widths above 16)
    {kFamTeam, 193, 5 * cr::kTeamWaves * cr::kWave, 1, kTeamPairLimit, kAnyLength, 32, "profiles/r01 (tools/calibrate_team_limit.py)"},
    // split by ROWS (cr_duo.h): 2 .. 8 waves per pair, all workgroups resident at once (wave limits: checked by fits)
    {kFamDuo, 257, kDuoMaxRows, kTeamPairLimit + 1, kMidWaveLimit / 2, kDuoMaxRows, 16, "profiles/r04/c3_share.txt, c3_share_lengths.txt"},
    // one wave per pair, pairs grouped by rows per lane (2 .. 5)
    {kFamSingle, 1, kAnyLength, 1, kAnyPairs, kAnyLength, 32, "profiles/r02 (tools/calibrate_rows_per_lane.py)"},
};

struct Layout {
    Family family = kFamSingle;
    int r_seed = 5, r_b = 5, wide_na = 0, wide_sync = 0;
    bool trio_few = false;
    bool ok = true;          // false: no family serves the list (tensors wider than 32 on a list the staged family cannot take)
};

// what the caller of cr_batch_set_pairs rules out (thread-local flags of the re-layouts)
struct LayoutMask {
    bool no_wide = false, no_trio = false, no_duo = false;
};

int launch_seed_team(int R, cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm);   // cr_dropins.h

// Rows per lane for a structure of n rows: the R in {2, 3, 4, 5} with the cheapest strips.  A strip walks all m
// columns; its measured cost per column (tools/calibrate_rows_per_lane.py, 4095 equal pairs per length, both
// kernels) is 1 : 1.175 : 1.534 : 1.77 for R = 2 : 3 : 4 : 5 -- not proportional to R, because the narrower kernels
// keep more waves per SIMD.  Ties go to the larger R.  300 rows -> 5 (one strip), 230 -> 4, 150 -> 3, 100 -> 2,
// 350 -> 3 (two strips), 450 -> 4 (two strips).
int rows_per_lane(int n) {
    if (g_cfg.force_r >= 2 && g_cfg.force_r <= 5) return g_cfg.force_r;       // calibration runs
    const int rs[4] = {5, 4, 3, 2};
    const double weight[4] = {1.77, 1.534, 1.175, 1.0};
    int best = 5;
    double best_cost = 1e300;
    for (int k = 0; k < 4; k++) {
        const double c = cr::strips_of(n, rs[k]) * weight[k];
        if (c < best_cost - 1e-9) {
            best_cost = c;
            best = rs[k];
        }
    }
    return best;
}

// Strip plans k_pair_duo is built for: (RA, RB) of cr_duo_instances.h, at most kDuoMaxWaves strips, columns resident in LDS
bool duo_fits(const StripPlan& p, int n_max, int m_max, int d_pad) {
    const int key = p.ra * 10 + p.rb;
    if (!(key == 11 || key == 21 || key == 22 || key == 32 || key == 33) || d_pad > 16) return false;
    if (p.ra != p.rb && (p.na < 1 || p.na >= cr::kDuoMaxWaves)) return false;
    const int waves = p.strips(n_max);
    if ((waves < 2 && !g_cfg.mid_any) || waves > cr::kDuoMaxWaves) return false;   // (CARETTA_MID_ANY: measurements)
    const size_t fill = cr::duo_sweep_lds_doubles<cr::kSwScore | cr::kDtw, cr::RbfCoords<1>>(waves, m_max);
    const size_t trace = cr::kExpDoubles + cr::trace_lds_doubles(1, n_max + m_max);
    return sizeof(double) * std::max(fill, trace) <= 64 * 1024;
}

// Can a pair list with these maxima run on the wide kernels with this strip plan?  (strips <= max_waves, the columns
// of the tensor sweep -- the larger of the two -- resident in LDS next to the edge rings)
bool wide_fits(const StripPlan& p, int n_max, int m_max, int d_pad, int max_waves = cr::kWideMaxWaves) {
    if (p.ra < 2 || p.ra > 3 || p.rb < 2 || p.rb > p.ra || d_pad > 16) return false;   // (the wide kernels are built for 2 or 3 rows per lane, widths up to 16)
    if (p.ra != p.rb && !(p.ra == 3 && p.rb == 2)) return false;                        // (the one mixed instance)
    const int waves = p.strips(n_max);
    if (waves > max_waves) return false;
    // (sized for sw_gap != 0, where the tensor sweep needs its columns resident too; the parameters come with cr_batch_run)
    const size_t seed = cr::kExpDoubles + (size_t)d_pad * m_max + (size_t)waves * (cr::kWideEdge + 8);
    const size_t align = cr::kExpDoubles + (size_t)3 * m_max + (size_t)waves * (3 * cr::kWideEdge + 8);
    const size_t trace = cr::kExpDoubles + cr::trace_team_lds_doubles(n_max + m_max);
    // (1 KB less than the CU's 160 KB: k_pair_wide also has a few hundred bytes of static LDS)
    return sizeof(double) * std::max(std::max(seed, align), trace) <= 159 * 1024;
}

// The strip plan of a wide launch.  A workgroup's waves are dealt round robin to the CU's four SIMDs; a SIMD issues one
// FP64-rate instruction per 4 cycles when two or more waves share it and a lone wave gets one per ~6.5,
// and all strips advance together (barriers), so a sweep step costs what the fullest SIMD needs for its row slots.  The
// skewed DTW fill takes lag * (S - 1) + m + 63 steps, the column sweeps of the seed and the score m + 16 * (S - 1).
// Candidates: 2 or 3 rows per lane everywhere, or 3 in the first nA strips and 2 in the others.
// 1200 rows: (3,3,3,2,2,2,2,2) -- 5,5,5,4 row slots per SIMD where seven strips of 3 have 6,6,6,3.
template <class Fits>
StripPlan choose_wide_plan(int n_max, int m_max, int sync_every, Fits fits) {
    StripPlan best{0, 0, 0};
    double best_cost = 1e300;
    auto consider = [&](const StripPlan& p) {
        if (!fits(p)) return;
        const int S = p.strips(n_max);
        int load[4] = {0, 0, 0, 0}, waves[4] = {0, 0, 0, 0};
        for (int w = 0; w < S; w++) {
            load[w & 3] += (p.ra == p.rb || w < p.na) ? p.ra : p.rb;
            waves[w & 3]++;
        }
        double step = 0.0;
        for (int k = 0; k < 4; k++) step = std::max(step, load[k] * (waves[k] >= 2 ? 4.0 : 6.5));
        const double lag = cr::kWave - 1 + sync_every;
        const double steps = (lag * (S - 1) + m_max + cr::kWave - 1) + 2.0 * (m_max + 16.0 * (S - 1));
        const double cost = step * steps;
        // (ties between mixed plans go to the one with more 3-row strips: 252 x 1200 x 1200 measured 2.207 / 2.162 / 2.175 ms
        // with nA = 4 against 2.213 / 2.183 / 2.189 with nA = 3 in three calibration runs)
        if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && p.ra != p.rb && best.ra != best.rb && p.na > best.na)) {
            best_cost = cost;
            best = p;
        }
    };
    consider(StripPlan{2, 2, 0});
    consider(StripPlan{3, 3, 0});
    for (int na = 1; na < cr::kWideMaxWaves; na++)
        if (na * cr::kWave * 3 < n_max) consider(StripPlan{3, 2, na});
    return best;
}

// the limits of a rule as the calibration switches move them
LayoutRule effective_rule(LayoutRule r) {
    const crcfg::Calibration& c = g_cfg;
    switch (r.family) {
        case kFamTrio:
            if (c.trio_pairs >= 0) r.pairs_hi = c.trio_pairs;
            if (c.trio_from >= 0) r.pairs_lo = c.trio_from + 1;
            if (c.trio_min_rows >= 0 && r.rows_lo == cr::kWave + 1) r.rows_lo = (int)c.trio_min_rows + 1;
            break;
        case kFamStaged:
            if (c.staged_waves >= 0) r.pairs_hi = kAnyPairs;
            if (c.staged_rows >= 0) r.rows_hi = (int)std::min<long long>(c.staged_rows, cr::kStagedMaxRows);
            break;
        case kFamWide:
        case kFamTeam:
            if (c.team_pairs >= 0) r.pairs_hi = c.team_pairs;
            break;
        case kFamDuo:
            if (c.team_pairs >= 0) r.pairs_lo = c.team_pairs + 1;
            if (c.mid_pairs >= 0) r.pairs_hi = c.mid_pairs;
            if (c.mid_any) r.rows_lo = 1;
            break;
        default: break;
    }
    return r;
}

// The kernel family (and its strip plan) for a list of `npairs` pairs whose longest structures have n_max rows / m_max columns.
Layout choose_layout(int n_max, int m_max, int d_pad, int64_t npairs, const LayoutMask mask) {
    const crcfg::Calibration& c = g_cfg;
    Layout out;
    out.r_seed = out.r_b = rows_per_lane(std::max(n_max, 1));
    if (npairs <= 0) return out;
    // calibration: CARETTA_WIDE="RA,RB,nA,B" forces the wide kernels with this plan
    if (c.wide.set && !mask.no_wide && c.wide.sync >= 1 && c.wide.sync <= cr::kWideMaxSync && c.wide.na >= 0 && c.wide.na < cr::kWideMaxWaves &&
        wide_fits(StripPlan{c.wide.ra, c.wide.rb, c.wide.ra == c.wide.rb ? 0 : c.wide.na}, n_max, m_max, d_pad)) {
        out.family = kFamWide;
        out.r_seed = c.wide.ra;
        out.r_b = c.wide.rb;
        out.wide_na = c.wide.ra == c.wide.rb ? 0 : c.wide.na;
        out.wide_sync = c.wide.sync;
        return out;
    }
    for (const LayoutRule& rule : kLayoutTable) {
        const LayoutRule r = effective_rule(rule);
        if (n_max < r.rows_lo || n_max > r.rows_hi || npairs < r.pairs_lo || npairs > r.pairs_hi || m_max > r.cols_hi || d_pad > r.d_pad_hi) continue;
        switch (r.family) {
            case kFamTrio: {
                if (!c.trio || c.no_team || mask.no_wide || mask.no_trio) break;
                out.family = kFamTrio;
                out.r_seed = out.r_b = std::max(2, (n_max + cr::kWave - 1) / cr::kWave);       // one strip of 2 .. 5 rows per lane
                out.trio_few = npairs <= kTeamPairLimit;
                return out;
            }
            case kFamStaged: {
                if (!c.staged || c.no_team || c.no_wide || c.wide.set || mask.no_wide) break;
                const cr::StagedShape shape = staged_shape(std::max(n_max, 1), std::max(m_max, 1));
                const int64_t wave_limit = c.staged_waves >= 0 ? c.staged_waves : kStagedWaveLimit;
                if (npairs * shape.waves > wave_limit) break;
                if ((double)npairs * (double)shape.pair_doubles() * sizeof(double) > 2.0 * 1024 * 1024 * 1024) break;
                // (the alignment columns of a pair and the term tile of the workgroup-wide sums share the LDS)
                if (sizeof(double) * ((size_t)cr::kExpDoubles + cr::trace_team_lds_doubles(n_max + m_max)) > 159 * 1024) break;
                out.family = kFamStaged;
                out.r_seed = out.r_b = shape.r;
                return out;
            }
            case kFamWide: {
                if (c.no_team || c.no_wide || mask.no_wide) break;
                const StripPlan p = choose_wide_plan(n_max, m_max, 8, [&](const StripPlan& q) { return wide_fits(q, n_max, m_max, d_pad); });
                if (!p.ra) break;
                out.family = kFamWide;
                out.r_seed = p.ra;
                out.r_b = p.rb;
                out.wide_na = p.na;
                out.wide_sync = 8;
                return out;
            }
            case kFamTeam: {
                if (c.no_team) break;
                out.family = kFamTeam;
                out.r_seed = out.r_b = (n_max + cr::kTeamWaves * cr::kWave - 1) / (cr::kTeamWaves * cr::kWave);
                return out;
            }
            case kFamDuo: {
                if (!c.mid || c.no_team || c.no_wide || mask.no_wide || mask.no_duo) break;
                StripPlan p{3, 2, 1};
                // up to 512 pairs of at most 320 rows: FOUR waves per pair (2 + 1 + 1 + 1 rows per lane: 2 048 waves still fit the
                // chip at once) -- 508 pairs of 300: 0.50 / 0.33 ms against 0.54 / 0.40 with two waves
                if (npairs <= 512 && n_max <= 5 * cr::kWave) p = StripPlan{2, 1, 1};
                // beyond the 1 088 rows that one strip of 3 and seven of 2 rows per lane cover: more strips of 3 (eight waves: up to 1 472 rows)
                if (n_max > 3 * cr::kWave + (cr::kDuoMaxWaves - 1) * 2 * cr::kWave)
                    p = StripPlan{3, 2, (n_max - cr::kDuoMaxWaves * 2 * cr::kWave + cr::kWave - 1) / cr::kWave};
                if (c.mid_plan.set) p = StripPlan{c.mid_plan.ra, c.mid_plan.rb, c.mid_plan.ra == c.mid_plan.rb ? 0 : c.mid_plan.na};
                const int64_t strips = std::max(p.strips(std::max(n_max, 1)), 1);
                // (every workgroup resident at once -- 16 waves per CU at <= 128 VGPRs --, and at most ~2.5 waves per SIMD for two
                // strips, 3 for more: beyond that the single-wave kernels fill the SIMDs by themselves)
                int64_t mid_limit = std::min<int64_t>(256 * (16 / strips), (strips == 2 ? kMidWaveLimit2 : kMidWaveLimit) / strips);
                // Long chains (seven or eight strips: 833 .. 1 472 rows) run in up to TWO rounds of 512 resident pairs: one wave per pair
                // is bound there by the latency of a wave that takes four or more strips in turn (9.4 ms for 1 200 rows whatever the
                // pair count), e.g. one GPU's share of BASELINE config 5 at 4 / 2 GPUs: 504 pairs of 1 200 9.6 -> 3.5 ms, 1 008 pairs
                // 9.4 -> 6.8; 900 rows 5.4 -> 2.4 / 5.4 -> 4.8; 600 rows (five strips) 1 008 pairs 2.4 -> 3.5: not those
                // (profiles/r05/long_share_layouts.txt)
                if (strips >= 7) mid_limit = 2 * 256 * 2;
                if (c.mid_pairs >= 0) mid_limit = c.mid_pairs;
                if (npairs > mid_limit || !duo_fits(p, n_max, m_max, d_pad)) break;
                out.family = kFamDuo;
                out.r_seed = p.ra;
                out.r_b = p.rb;
                out.wide_na = p.na;
                out.wide_sync = 8;
                return out;
            }
            case kFamSingle: return out;
        }
    }
    out.ok = d_pad <= 32;        // (every rule but the staged one ends at width 32)
    return out;
}

// Size class of a pair of n rows and m columns: A = one strip of the single-strip families (<= 320 rows; the split by function
// keeps up to 1 280 columns resident), B = the row-split families (<= 1 472 rows and columns), C = everything else.
int size_class(int n, int m) {
    if (n <= 5 * cr::kWave && m <= kMidMaxColumns) return 0;
    return (n <= kDuoMaxRows && m <= kDuoMaxRows) ? 1 : 2;
}

// What cr_batch_set_pairs decides about a pair list before anything touches the device: the longest rows / columns, the pairs
// per size class, and whether the list is split into classes.
//
// Size classes.  The layout of a list follows from its LONGEST structure, so one 600-residue member moves a family of
// 150-residue structures to another kernel family (or out of every family built for its size).  A ragged list of at most
// kClassSplitPairs pairs -- more fill the chip one wave per pair, which groups by rows per lane already -- is therefore split
// into at most three classes by rows (<= 320 / <= 1 472 / longer; size_class()), each laid out as a list of its own -- when
// that gives any class another family than one wave per pair and the classes do not all agree with the whole list's family.
struct ListPlan {
    int n_max = 0, m_max = 0;
    int64_t in_class[3] = {0, 0, 0};
    int cn[3] = {0, 0, 0}, cm[3] = {0, 0, 0};
    bool split = false;
    Layout whole;                 // the layout of the list as ONE list
    Layout of_class[3];           // (split) the layouts of the classes
};

int plan_list(const int64_t* offsets, int64_t P, int d_pad, const int32_t* pairs, int64_t npairs, const LayoutMask mask, bool may_split,
              ListPlan& out) {
    out = ListPlan{};
    for (int64_t p = 0; p < npairs; p++) {
        const int64_t i = pairs[2 * p], j = pairs[2 * p + 1];
        CR_REQUIRE(i >= 0 && i < P && j >= 0 && j < P, "pair index out of range");
        const int n = (int)(offsets[i + 1] - offsets[i]), m = (int)(offsets[j + 1] - offsets[j]);
        out.n_max = std::max(out.n_max, n);
        out.m_max = std::max(out.m_max, m);
        const int c = size_class(n, m);
        out.in_class[c]++;
        out.cn[c] = std::max(out.cn[c], n);
        out.cm[c] = std::max(out.cm[c], m);
    }
    out.whole = choose_layout(out.n_max, out.m_max, d_pad, npairs, mask);
    const int nclasses = (out.in_class[0] > 0) + (out.in_class[1] > 0) + (out.in_class[2] > 0);
    if (g_cfg.classes && nclasses >= 2 && npairs <= kClassSplitPairs && may_split && d_pad <= 32) {
        bool all_same = true, any_special = false;
        for (int c = 0; c < 3; c++) {
            if (!out.in_class[c]) continue;
            out.of_class[c] = choose_layout(out.cn[c], out.cm[c], d_pad, out.in_class[c], mask);
            all_same = all_same && out.of_class[c].family == out.whole.family;
            any_special = any_special || out.of_class[c].family != kFamSingle;
        }
        out.split = any_special && !all_same;
    }
    return CR_OK;
}

int public_family(const Layout& l) {
    switch (l.family) {
        case kFamTeam: return CR_LAYOUT_TEAM;
        case kFamWide: return CR_LAYOUT_WIDE;
        case kFamDuo: return CR_LAYOUT_DUO;
        case kFamTrio: return CR_LAYOUT_TRIO;
        case kFamStaged: return CR_LAYOUT_STAGED;
        default: return CR_LAYOUT_SINGLE;
    }
}

