// The schedule of cbo_gp_fit_sweep, chosen from this device's own timings.  Host logic only (no HIP): included by
// cbo_api.hip, and compiled on its own by tests/test_host_logic.py against a simulated device.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <utility>

// How many panel pairs of the overlapped refit + sweep go through the right-looking pipeline (under the factorisation, on
// the CUs its chain leaves idle) before the rest is left to one left-looking launch; whether the pipeline's updates go
// in groups of two pairs (K = 512); whether to overlap at all.  The best answer depends on how long this device's chain
// takes per panel, how fast its strips go and how the two slow each other down when they share CUs -- rounds 2 and 3
// fitted a budget curve through three measured shapes (and a row cut-off, and a grouping rule).  Now every shape
// (padded rows, padded candidates) of a context is measured on the calls the caller makes anyway:
//   1. the first call of a shape runs the analytic split: the largest number of pairs whose pipeline -- its share of the
//      sweep's strip stages, on the CUs the sweep streams may use -- ends no later than the factorisation alone would, at
//      the rates round 4 measured on this device (schedule_default_rates); everything pipelined when the strips cannot
//      fill the device.  (Round 4 timed the plain sequence twice first: a fresh context's first calls, and every call of a
//      caller whose calls are never sampled, then ran fit-then-sweep.)
//   2. whenever the climb measures the plain sequence itself, the times of its halves refine those rates;
//   3. from there the split climbs on the measured time of the calls themselves: neighbours first, doubling steps while
//      they pay, single steps to finish; the candidates are the plain sequence, the overlapped call with an empty pipeline,
//      and the multiples of a pair (or of a group of pairs) up to everything pipelined.  A candidate's time is the
//      smaller of two calls, the call after a change of schedule not counted (it pays for the change); two candidates
//      within 1 % of each other get two more calls each before the decision.  Once settled, the other grouping is tried
//      at the settled split and kept if it is faster.
// Results do not depend on the schedule (same bits whatever the split: tests/test_parity_gpu.py), so the exploration
// only costs the few per cent by which a neighbouring split is slower, on the first ~20-30 calls of a shape.  A new
// shape near a settled one (the model grew by a panel, a grid of another size) starts from that one's split.  CBO_HIP_SCHEDULE_TUNE=0 stops
// after step 2; CBO_HIP_PIPE_TAIL / CBO_HIP_PIPE_GROUP / CBO_HIP_OVERLAP force a schedule as before (the tuner is
// bypassed); cbo_schedule_report prints what was measured and chosen.

// a candidate's calls: its time is the median of them (the mean of the middle two) -- an estimate that does not improve
// with the number of calls, as the smallest of them would
struct ScheduleSample {
    double t[8]; int count = 0, need = 2;
    void add(double ms)
    {
        if (count < 8) t[count++] = ms;
        else { int worst = 0; for (int i = 1; i < 8; ++i) if (t[i] > t[worst]) worst = i; if (ms < t[worst]) t[worst] = ms; }
    }
    double ms() const
    {
        if (count == 0) return 1e300;
        double v[8];
        for (int i = 0; i < count; ++i) v[i] = t[i];
        for (int i = 1; i < count; ++i) { const double x = v[i]; int j = i - 1; while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; } v[j + 1] = x; }
        return 0.5 * (v[(count - 1) / 2] + v[count / 2]);
    }
};
constexpr int kSequence = -1;              // the "number of pairs" that stands for the plain sequence (fit, then sweep)
struct ScheduleEntry {
    enum State { COLD, SEQUENCE, BASE, NEIGHBOURS, CLIMB, GROUPING, SETTLED, RENEW };
    State state = COLD;
    int64_t n_pad = 0;
    int all_pairs = 0, strips = 0;
    int group = 0;                         // 0 = updates pair by pair, 2 = in groups of two pairs
    int cur = kSequence;                   // pairs in force; 0 = overlapped with an empty pipeline; kSequence = the plain sequence
    int probe = kSequence, probe_group = 0;   // what the next call runs (states BASE .. GROUPING)
    int dir = 0, step = 0;
    int up = kSequence, down = kSequence;  // NEIGHBOURS: the two candidates (== cur: none on that side)
    int alt_group = 0, alt_pairs = 0;      // GROUPING: the candidate with the other grouping
    int calls = 0;
    int last_pairs = kSequence - 1, last_group = 0;   // the previous call's schedule (a call after a change is not sampled)
    bool group_tried = false, relooked = false;
    int first_pairs = kSequence - 1, first_group = 0;   // the first split: measured on the context's first calls of the shape
    bool first_renewed = false;            // ... and measured again before it is allowed to lose (schedule_settle)
    int retries = -1;                      // jitter retries of the shape's calls (calls with another count are not sampled)
    int retry_mismatch = 0;                // SETTLED: consecutive calls that needed another number of retries than e.retries
    uint64_t last_use = 0;                 // the table's call counter when the shape was last asked for (eviction)
    int ran_pairs = kSequence, ran_group = 0;   // what the shape's last call ran (cbo_schedule_report: tests assert it)
    double fact_alone_us = 0.0, sweep_alone_us = 0.0;
    std::map<std::pair<int, int>, ScheduleSample> samples;   // (group, pairs) -> fastest of its calls
};

typedef std::map<std::pair<int64_t, int64_t>, ScheduleEntry> ScheduleTable;   // (padded rows, padded candidates) ->

struct ScheduleChoice { int pairs = kSequence, group = 0; bool sample = false; };
constexpr double kScheduleGain = 0.003;    // a candidate replaces the current one when it is this much faster
constexpr double kScheduleClose = 0.01;    // two candidates closer than this are sampled twice more before the decision
constexpr int kScheduleMaxCalls = 96;      // exploration gives up (settles where it is) after this many calls of a shape
constexpr int kScheduleMaxShapes = 64;     // shapes a context keeps; beyond it the one unused for longest goes
constexpr int kScheduleRetryDrift = 3;     // a settled shape is measured afresh after this many consecutive calls whose
                                           // jitter retries differ from the ones it was measured with

static inline bool schedule_tune_enabled()
{
    static const bool on = [] { const char *e = std::getenv("CBO_HIP_SCHEDULE_TUNE"); return !(e && std::atoi(e) == 0); }();
    return on;
}
// The candidates along the axis, in order: kSequence, 0 (overlapped, empty pipeline), every number of pairs up to all_pairs.
// (Until round 5 a grouped schedule only took whole groups; pairs % group pairs now go alone AHEAD of the first group --
// SweepPipe::lead -- so that the bulk stream has work as soon as the first pair is solved: every count is a candidate.)
static inline bool schedule_is_candidate(const ScheduleEntry &e, int group, int p)
{
    (void)group;
    return p == kSequence || (p >= 0 && p <= e.all_pairs);
}
// the nearest candidate at or beyond p in direction dir
static inline int schedule_snap(const ScheduleEntry &e, int group, int p, int dir)
{
    if (p < 0) return kSequence;
    if (p >= e.all_pairs) return e.all_pairs;
    while (!schedule_is_candidate(e, group, p)) p += dir > 0 ? 1 : -1;
    return p;
}
// `steps` candidates away from p (a candidate) in direction dir, clamped at the ends of the axis; a step down of more
// than one candidate stops at the empty pipeline (the plain sequence is only ever reached from there)
static inline int schedule_next(const ScheduleEntry &e, int group, int p, int dir, int steps)
{
    p = schedule_snap(e, group, p, -1);
    for (int k = 0; k < steps; ++k) {
        if (dir > 0) {
            if (p >= e.all_pairs) break;
            p = schedule_snap(e, group, p + 1, +1);
        } else {
            if (p == kSequence || (p == 0 && steps > 1)) break;
            p = p == 0 ? kSequence : schedule_snap(e, group, p - 1, -1);
        }
    }
    return p;
}
static inline std::pair<int, int> schedule_key(int group, int pairs) { return {pairs <= 0 ? 0 : group, pairs}; }   // (no updates: no grouping)
static inline double schedule_ms(const ScheduleEntry &e, int group, int pairs)
{
    auto it = e.samples.find(schedule_key(group, pairs));
    return (it == e.samples.end() || it->second.count < it->second.need) ? 1e300 : it->second.ms();
}
static inline void schedule_probe(ScheduleEntry &e, int group, int pairs) { e.probe = pairs; e.probe_group = pairs < 0 ? e.group : group; }
// two measured candidates too close to call: both are sampled twice more (once), the caller comes back
static inline bool schedule_close_call(ScheduleEntry &e, int ga, int pa, int gb, int pb)
{
    const double ta = schedule_ms(e, ga, pa), tb = schedule_ms(e, gb, pb);
    if (ta > 1e299 || tb > 1e299 || std::fabs(ta - tb) > kScheduleClose * tb) return false;
    ScheduleSample &sa = e.samples[schedule_key(ga, pa)], &sb = e.samples[schedule_key(gb, pb)];
    if (sa.need >= 4 && sb.need >= 4) return false;
    sa.need = 4; sb.need = 4;
    if (sa.count < sa.need) schedule_probe(e, ga, pa); else schedule_probe(e, gb, pb);
    return true;
}
// strip stages (32 rows of one 64-column strip) of the pipeline's first `pairs` pairs; the whole sweep has 2 nb (nb + 1)
static inline double pipeline_stages(int nb, int pairs)
{
    double s = 0.0;
    for (int p = 0; p < pairs; ++p) {
        const int rows_below = nb - 2 * (p + 1);
        s += 12.0 + 8.0 * (rows_below > 0 ? rows_below : 0);
    }
    return s;
}
static inline int schedule_first_split(int n_cu, int n_cu_pipe, const ScheduleEntry &e)
{
    const int nb = (int)(e.n_pad / 128);
    if (e.strips < n_cu_pipe) return e.all_pairs;             // the strip kernel could not fill the device on its own
    const double total = 2.0 * nb * (nb + 1);
    int pairs = 0;
    while (pairs < e.all_pairs) {
        const double pipe_us = pipeline_stages(nb, pairs + 1) / total * e.sweep_alone_us * (double)n_cu / n_cu_pipe;
        if (pipe_us > e.fact_alone_us) break;
        ++pairs;
    }
    return pairs;
}
// Rates of the two halves of the plain sequence for a shape nothing has been measured on yet (round-4 measurements on
// MI355X, DESIGN.md: the factorisation alone 54 / 92 / 232 us per 128-row panel at 4096 / 8192 / 16384 rows, 37 at 1024; the
// strip kernel 2.2 us per 32-row stage and round of strips): what a call that may not be sampled -- it runs under the
// library's profiling timers -- uses for its first split on a COLD shape instead of falling back to the plain sequence.
static inline void schedule_default_rates(int n_cu, ScheduleEntry &e)
{
    const double n = (double)e.n_pad;
    const double per_panel = n <= 1024.0 ? 37.0
                             : n <= 4096.0 ? 37.0 + (54.0 - 37.0) * (n - 1024.0) / 3072.0
                             : n <= 8192.0 ? 54.0 + (92.0 - 54.0) * (n - 4096.0) / 4096.0
                                           : 92.0 + (232.0 - 92.0) * (n - 8192.0) / 8192.0;
    const int nb = (int)(e.n_pad / 128);
    const int rounds = (e.strips + n_cu - 1) / (n_cu > 0 ? n_cu : 1);
    e.fact_alone_us = per_panel * nb;
    e.sweep_alone_us = 2.2 * 2.0 * nb * (nb + 1) * (rounds > 0 ? rounds : 1);
}
static inline int schedule_first_split(int n_cu, int n_cu_pipe, const ScheduleEntry &e);
static inline int schedule_snap(const ScheduleEntry &e, int group, int p, int dir);
// the split of a call that cannot take part in the measurement, on a shape that has not been measured: analytic
static inline ScheduleChoice schedule_static_choice(int n_cu, int n_cu_pipe, const ScheduleEntry &e)
{
    ScheduleEntry d = e;
    if (d.fact_alone_us <= 0.0 || d.sweep_alone_us <= 0.0) schedule_default_rates(n_cu, d);
    ScheduleChoice ch;
    ch.group = e.group;
    ch.pairs = schedule_snap(e, e.group, schedule_first_split(n_cu, n_cu_pipe, d), -1);
    ch.sample = false;
    return ch;
}

static inline ScheduleEntry &schedule_entry(ScheduleTable &table, int n_cu_pipe, int64_t n_pad, int64_t strips, int64_t m_pad)
{
    auto key = std::make_pair(n_pad, m_pad);
    uint64_t now = 0;
    for (const auto &kv : table) if (kv.second.last_use > now) now = kv.second.last_use;
    ++now;
    auto it = table.find(key);
    if (it != table.end()) { it->second.last_use = now; return it->second; }
    if ((int)table.size() >= kScheduleMaxShapes) {          // a caller whose shapes never repeat: the table stays bounded
        auto oldest = table.begin();
        for (auto jt = table.begin(); jt != table.end(); ++jt) if (jt->second.last_use < oldest->second.last_use) oldest = jt;
        table.erase(oldest);
    }
    ScheduleEntry e;
    const int nb = (int)(n_pad / 128);
    e.n_pad = n_pad;
    e.last_use = now;
    e.all_pairs = (nb + 1) / 2;
    e.strips = (int)strips;
    // (a full round of strips: the bulk stream bounds the pipeline, its updates go in groups -- to begin with)
    e.group = (e.strips >= n_cu_pipe && e.all_pairs >= 4) ? 2 : 0;
    // A settled shape nearby (rows and candidates within a factor of two each, on the same side of "a full round of
    // strips"): start from its split, as a fraction of the pairs, and its grouping -- a model that grew by a panel, a grid
    // of another size.  A caller whose shapes never repeat then still runs a schedule measured next door instead of the
    // plain sequence every time; only a context's first shape (and one far from everything seen) pays the two timing calls.
    const ScheduleEntry *seed = nullptr;
    double seed_dist = 1e300;
    for (const auto &kv : table) {
        const ScheduleEntry &o = kv.second;
        if (o.state != ScheduleEntry::SETTLED || o.all_pairs < 1 || (o.strips >= n_cu_pipe) != (e.strips >= n_cu_pipe)) continue;
        const double dist = std::fabs(std::log2((double)o.n_pad / (double)n_pad)) +
                            std::fabs(std::log2((double)(o.strips > 0 ? o.strips : 1) / (double)(e.strips > 0 ? e.strips : 1)));
        if (dist < seed_dist) { seed_dist = dist; seed = &o; }
    }
    if (seed && std::fabs(std::log2((double)seed->n_pad / (double)n_pad)) <= 1.0 &&
        std::fabs(std::log2((double)(seed->strips > 0 ? seed->strips : 1) / (double)(e.strips > 0 ? e.strips : 1))) <= 1.0) {
        if (e.all_pairs >= 4) e.group = seed->group;
        e.cur = seed->cur <= 0 ? seed->cur
                               : schedule_snap(e, e.group, (int)((double)seed->cur / seed->all_pairs * e.all_pairs + 0.5), -1);
        e.retries = -1;
        e.state = ScheduleEntry::BASE;
        schedule_probe(e, e.group, e.cur);
    }
    return table.emplace(key, e).first->second;
}

// what this call runs.  may_sample: the call's wall time means something (it does not run under the library's profiling
// timers, which serialise the streams).  Whether the caller asks for the per-candidate outputs does not matter: every
// candidate schedule of the shape pays the same copies.  A call that may not be sampled runs what is in force -- and on a
// shape nothing is known about yet, the analytic first split (everything pipelined when the strips cannot fill the device),
// not the plain sequence: a caller that is never sampled must not be left with fit-then-sweep for ever.
static inline ScheduleChoice schedule_choose(ScheduleEntry &e, bool may_sample, int n_cu = 256, int n_cu_pipe = 224)
{
    ScheduleChoice ch;
    const bool exploring = may_sample && e.state != ScheduleEntry::SETTLED;
    if (e.state == ScheduleEntry::COLD) {
        // nothing is known about the shape: the analytic split (round 5; round 4 ran the plain sequence twice here to time its
        // halves, which left a fresh context's first calls -- and every call of a caller that is never sampled -- with
        // fit-then-sweep).  A sampled call then starts the climb from it.
        ch = schedule_static_choice(n_cu, n_cu_pipe, e);
        ch.sample = may_sample;
        return ch;
    }
    else if (exploring) { ch.pairs = e.probe; ch.group = e.probe_group; }
    else { ch.pairs = e.cur; ch.group = e.group; }
    ch.sample = exploring;
    return ch;
}

static inline void schedule_look_around(ScheduleEntry &e);
// The climb is local: what it settles on is the fastest of everything it has measured -- the plain sequence included (a
// shape whose fits need jitter retries repeats its pipeline with every retry: the sequence wins by far, nowhere near the
// first split).  When that is not where the climb stands, its neighbours get one look (once).
static inline void schedule_settle(ScheduleEntry &e)
{
    double best = schedule_ms(e, e.group, e.cur);
    bool moved = false;
    for (const auto &kv : e.samples) {
        if (kv.second.count < kv.second.need) continue;
        const double t = kv.second.ms();
        if (t < best * (1.0 - kScheduleGain)) {
            best = t; e.cur = kv.first.second; moved = true;
            if (e.cur > 0) e.group = kv.first.first;
        }
    }
    // The first split was timed on the first calls of the shape -- in a fresh context those are the device's first work after
    // idling, clocks still rising: a trace of the 2048 x 16384 shape had it at 1.81 / 1.77 ms on calls 2-3 where the same
    // schedule runs at 1.58 ms once warm, and candidates measured ten calls later, at 1.71 ms, beat it (the schedule scan's
    // outlier of rounds 4-5).  Before the first split is allowed to lose it is measured once more.
    if (!e.first_renewed && e.state != ScheduleEntry::SETTLED && e.calls <= kScheduleMaxCalls && e.first_pairs >= 0 &&
        (e.cur != e.first_pairs || (e.cur > 0 && e.group != e.first_group))) {
        e.first_renewed = true;
        e.samples.erase(schedule_key(e.first_group, e.first_pairs));
        e.state = ScheduleEntry::RENEW;
        schedule_probe(e, e.first_group, e.first_pairs);
        return;
    }
    if (moved && !e.relooked && e.state != ScheduleEntry::SETTLED && e.calls <= kScheduleMaxCalls) {
        e.relooked = true;
        schedule_look_around(e);
        return;
    }
    e.state = ScheduleEntry::SETTLED;
    schedule_probe(e, e.group, e.cur);
}

// after the climb: the other grouping at the settled split, once
static inline void schedule_try_grouping(ScheduleEntry &e)
{
    if (e.group_tried || e.all_pairs < 4 || e.cur < 2) { schedule_settle(e); return; }
    e.group_tried = true;
    const int other = e.group >= 2 ? 0 : 2;
    const int at = schedule_snap(e, other, e.cur, -1);
    if (at <= 0) { schedule_settle(e); return; }
    e.alt_group = other; e.alt_pairs = at;
    schedule_probe(e, other, at);
    e.state = ScheduleEntry::GROUPING;
}

// the climb: from cur in direction dir, `step` candidates at a time, doubling while it pays, single steps to finish
static inline void schedule_climb(ScheduleEntry &e)
{
    e.state = ScheduleEntry::CLIMB;
    for (int guard = 0; guard < 64; ++guard) {
        const int cand = schedule_next(e, e.group, e.cur, e.dir, e.step);
        if (cand == e.cur) {
            if (e.step > 1) { e.step = 1; continue; }
            schedule_look_around(e);
            return;
        }
        const double t = schedule_ms(e, e.group, cand);
        if (t > 1e299) { schedule_probe(e, e.group, cand); return; }
        if (schedule_ms(e, e.group, e.cur) > 1e299) { schedule_probe(e, e.group, e.cur); return; }
        if (schedule_close_call(e, e.group, cand, e.group, e.cur)) return;
        if (t < schedule_ms(e, e.group, e.cur) * (1.0 - kScheduleGain)) { e.cur = cand; e.step *= 2; }
        else if (e.step > 1) e.step = 1;
        else { schedule_look_around(e); return; }
    }
    schedule_settle(e);
}

// cur and its two neighbours: stay (then the other grouping), or set off in the better direction
static inline void schedule_neighbours(ScheduleEntry &e)
{
    e.state = ScheduleEntry::NEIGHBOURS;
    if (schedule_ms(e, e.group, e.cur) > 1e299) { schedule_probe(e, e.group, e.cur); return; }
    if (e.up != e.cur && schedule_ms(e, e.group, e.up) > 1e299) { schedule_probe(e, e.group, e.up); return; }
    if (e.down != e.cur && schedule_ms(e, e.group, e.down) > 1e299) { schedule_probe(e, e.group, e.down); return; }
    const double tu = e.up != e.cur ? schedule_ms(e, e.group, e.up) : 1e300;
    const double td = e.down != e.cur ? schedule_ms(e, e.group, e.down) : 1e300;
    const int best = tu < td ? e.up : e.down;
    if (best == e.cur) { schedule_settle(e); return; }                       // (a one-candidate axis)
    if (schedule_close_call(e, e.group, best, e.group, e.cur)) return;
    if (!((tu < td ? tu : td) < schedule_ms(e, e.group, e.cur) * (1.0 - kScheduleGain))) { schedule_try_grouping(e); return; }
    e.dir = tu < td ? +1 : -1;
    e.cur = best;
    e.step = 2;
    schedule_climb(e);
}

// the climb has ended: a doubled step may have carried it across the best candidate -- both neighbours of where it
// stands once more (every move is to a strictly faster candidate: this ends)
static inline void schedule_look_around(ScheduleEntry &e)
{
    e.up = schedule_next(e, e.group, e.cur, +1, 1);
    e.down = schedule_next(e, e.group, e.cur, -1, 1);
    schedule_neighbours(e);
}

// a call has ended: its time, and (the plain sequence) the times of its halves
// `retries`: the jitter retries the call needed (every retry repeats the factorisation -- and, overlapped, the pipeline:
// calls are comparable when they needed the same number; the shape's first sampled call sets it), < 0: not to be sampled
static inline void schedule_report(int n_cu, int n_cu_pipe, ScheduleEntry &e, const ScheduleChoice &ch, int retries, double ms,
                            double fact_us, double sweep_us)
{
    if (e.state == ScheduleEntry::SETTLED) {
        // The schedule was measured at e.retries jitter retries per fit; a pipelined split repeats its pipeline with every
        // retry.  When the data drift so that the fits keep needing another number, what was measured no longer describes
        // the shape: start over (the next sampled calls time the plain sequence again).
        if (retries >= 0 && e.retries >= 0 && retries != e.retries) {
            if (++e.retry_mismatch >= kScheduleRetryDrift) {
                const ScheduleEntry fresh;
                e.state = ScheduleEntry::COLD; e.samples.clear(); e.calls = 0; e.retries = -1; e.retry_mismatch = 0;
                e.cur = fresh.cur; e.probe = fresh.probe; e.probe_group = 0; e.group_tried = false; e.relooked = false;
                e.first_pairs = fresh.first_pairs; e.first_group = 0; e.first_renewed = false;
                e.fact_alone_us = 0.0; e.sweep_alone_us = 0.0; e.last_pairs = fresh.last_pairs;
            }
        } else if (retries >= 0) e.retry_mismatch = 0;
        return;
    }
    if (!ch.sample) { e.last_pairs = kSequence - 1; return; }    // (an unsampled call in between: the next one is a change)
    ++e.calls;
    const bool changed = ch.pairs != e.last_pairs || (ch.pairs > 0 && ch.group != e.last_group);
    e.last_pairs = ch.pairs; e.last_group = ch.group;
    if (e.state == ScheduleEntry::COLD) {                        // (allocations, code loading: the call itself is not sampled)
        e.cur = ch.pairs;
        if (ch.pairs > 0) e.group = ch.group;
        e.first_pairs = ch.pairs; e.first_group = ch.pairs > 0 ? ch.group : 0;
        if (!schedule_tune_enabled()) { e.state = ScheduleEntry::SETTLED; schedule_probe(e, e.group, e.cur); return; }
        e.state = ScheduleEntry::BASE;
        schedule_probe(e, e.group, e.cur);
        return;
    }
    if (e.calls > kScheduleMaxCalls) { schedule_settle(e); return; }
    if (changed || retries < 0) return;    // the first call of a schedule pays for the change
    if (e.retries < 0) e.retries = retries;
    if (retries != e.retries) return;
    ScheduleSample &sm = e.samples[schedule_key(ch.group, ch.pairs)];
    sm.add(ms);
    // two calls that disagree by more than 3 % (a hiccup of the host or the clocks in one of them): a third decides -- the
    // median of three ignores one outlier, the mean of two does not (round 5: a 12 % outlier in one of two samples left the
    // 2048 x 16384 shape a candidate short of its best split)
    if (sm.count == 2 && sm.need == 2 && std::fabs(sm.t[0] - sm.t[1]) > 0.03 * (sm.t[0] < sm.t[1] ? sm.t[0] : sm.t[1])) sm.need = 3;
    if (ch.pairs == kSequence && fact_us > 0.0) {
        if (e.fact_alone_us == 0.0 || fact_us < e.fact_alone_us) e.fact_alone_us = fact_us;
        if (e.sweep_alone_us == 0.0 || sweep_us < e.sweep_alone_us) e.sweep_alone_us = sweep_us;
    }
    if (sm.count < sm.need) return;                              // the same candidate once more
    switch (e.state) {
        case ScheduleEntry::SEQUENCE:
            e.cur = schedule_snap(e, e.group, schedule_first_split(n_cu, n_cu_pipe, e), -1);
            if (!schedule_tune_enabled()) { schedule_settle(e); return; }
            e.state = ScheduleEntry::BASE;
            schedule_probe(e, e.group, e.cur);
            return;
        case ScheduleEntry::BASE:
            if (!schedule_tune_enabled()) { schedule_settle(e); return; }
            // the plain sequence is measured once, right behind the first split: what the climb settles on is the fastest of
            // everything measured, and where the sequence wins by far (fits that retry with jitter repeat a pipelined
            // split's pipeline every time) the climb starts next to it instead of walking down to it
            if (schedule_ms(e, 0, kSequence) > 1e299) { schedule_probe(e, e.group, kSequence); return; }
            if (schedule_ms(e, e.group, e.cur) > 1e299) { schedule_probe(e, e.group, e.cur); return; }
            if (e.cur > 0 && schedule_ms(e, 0, kSequence) < 0.7 * schedule_ms(e, e.group, e.cur)) e.cur = 0;
            e.up = schedule_next(e, e.group, e.cur, +1, 1);
            e.down = schedule_next(e, e.group, e.cur, -1, 1);
            schedule_neighbours(e);
            return;
        case ScheduleEntry::NEIGHBOURS:
            schedule_neighbours(e);
            return;
        case ScheduleEntry::RENEW:                               // the first split's second measurement is in: decide
            schedule_settle(e);
            return;
        case ScheduleEntry::CLIMB:
            schedule_climb(e);
            return;
        case ScheduleEntry::GROUPING:
            if (schedule_ms(e, e.alt_group, e.alt_pairs) > 1e299) { schedule_probe(e, e.alt_group, e.alt_pairs); return; }
            if (schedule_ms(e, e.group, e.cur) > 1e299) { schedule_probe(e, e.group, e.cur); return; }
            if (schedule_close_call(e, e.alt_group, e.alt_pairs, e.group, e.cur)) return;
            if (schedule_ms(e, e.alt_group, e.alt_pairs) < schedule_ms(e, e.group, e.cur) * (1.0 - kScheduleGain)) {
                e.group = e.alt_group;                           // adopted: its neighbours once more
                e.cur = e.alt_pairs;
                e.up = schedule_next(e, e.group, e.cur, +1, 1);
                e.down = schedule_next(e, e.group, e.cur, -1, 1);
                schedule_neighbours(e);
                return;
            }
            schedule_settle(e);
            return;
        default: return;
    }
}

