// Drives cbo_with_oop_amd/csrc/schedule_tuner.h against a simulated device: a step-time curve over the candidates (the
// plain sequence, the empty pipeline, pairs), multiplicative noise, a penalty on the call after a change of schedule.
// usage: schedule_tuner_sim n_pad m_pad n_cu n_cu_pipe curve noise seed      (tests/test_host_logic.py)
// prints: "<calls> <pairs> <group> <settled 0/1> <ms of the choice> <ms of the best candidate>"
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <random>
#include <string>
#include "schedule_tuner.h"

static double step_ms(int curve, const ScheduleEntry &e, int group, int pairs)
{
    const double P = e.all_pairs;
    const double g = (pairs > 0 && group >= 2) ? 0.99 : 1.0;              // grouped updates: 1 % faster
    switch (curve) {
        case 0: {                                                        // a valley at a quarter of the pairs
            if (pairs < 0) return 6.5;
            const double x = pairs / P - 0.25;
            return (5.4 + 8.0 * x * x) * g;
        }
        case 1: return pairs < 0 ? 20.3 : (19.9 + 0.8 * pairs) * g;       // the empty pipeline is best
        case 2: return pairs < 0 ? 12.2 : (22.7 - 12.9 * pairs / P) * g;  // everything pipelined is best, the sequence second
        case 3: return pairs < 0 ? 100.0 : (101.0 + 0.5 * pairs) * g;     // the plain sequence is best
        case 5: return pairs < 0 ? 197.0 : (330.0 + 2.0 * std::fabs(pairs - 12.0)) * g;   // retries: the pipeline repeats, the sequence wins by far
        default: {                                                       // flat with a far valley (at 3/8)
            if (pairs < 0) return 24.6;
            const double x = pairs / P - 0.375;
            return (22.5 + 30.0 * x * x) * g;
        }
    }
}

// "unsampled n_pad m_pad n_cu n_cu_pipe": a caller none of whose calls may be sampled (it runs under the library's profiling
// timers) on a shape nothing is known about: every call must run the analytic split, never the plain sequence; prints
// "<pairs> <group> <all_pairs> <state>" of the 1st and the 30th call.
// "drift ...": a settled shape whose fits start needing another number of jitter retries is measured afresh after
// kScheduleRetryDrift such calls in a row; prints "<state before> <state after two> <state after three> <state after a sampled call>".
// "bounded": more shapes than kScheduleMaxShapes leave the table at its bound; prints the table's size.
static int special(int argc, char **argv)
{
    const std::string mode = argv[1];
    if (mode == "bounded") {
        ScheduleTable table;
        for (int i = 0; i < 3 * kScheduleMaxShapes; ++i) schedule_entry(table, 224, 1024 + 128 * i, 256, 16384);
        std::printf("%d\n", (int)table.size());
        return 0;
    }
    if (argc < 6) return 2;
    const int64_t n_pad = std::atoll(argv[2]), m_pad = std::atoll(argv[3]);
    const int n_cu = std::atoi(argv[4]), n_cu_pipe = std::atoi(argv[5]);
    ScheduleTable table;
    ScheduleEntry &e = schedule_entry(table, n_cu_pipe, n_pad, m_pad / 64, m_pad);
    if (mode == "unsampled") {
        for (int call = 0; call < 30; ++call) {
            const ScheduleChoice ch = schedule_choose(e, false, n_cu, n_cu_pipe);
            if (call == 0 || call == 29) std::printf("%d %d %d %d\n", ch.pairs, ch.group, e.all_pairs, (int)e.state);
            schedule_report(n_cu, n_cu_pipe, e, ch, 0, 5.0, 0.0, 0.0);
        }
        return 0;
    }
    if (mode == "drift") {
        int calls = 0;
        while (e.state != ScheduleEntry::SETTLED && calls++ < 400) {
            const ScheduleChoice ch = schedule_choose(e, true, n_cu, n_cu_pipe);
            schedule_report(n_cu, n_cu_pipe, e, ch, 0, step_ms(0, e, ch.group, ch.pairs), ch.pairs < 0 ? 1700.0 : 0.0, ch.pairs < 0 ? 5100.0 : 0.0);
        }
        const int s0 = (int)e.state;
        int st[3];
        for (int k = 0; k < 3; ++k) {
            const ScheduleChoice ch = schedule_choose(e, true, n_cu, n_cu_pipe);
            schedule_report(n_cu, n_cu_pipe, e, ch, 1, 9.0, 0.0, 0.0);            // the fits now retry once
            st[k] = (int)e.state;
        }
        const ScheduleChoice ch = schedule_choose(e, true, n_cu, n_cu_pipe);
        std::printf("%d %d %d %d\n", s0, st[1], st[2], ch.pairs);
        return 0;
    }
    return 2;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && (argv[1][0] < '0' || argv[1][0] > '9')) return special(argc, argv);
    if (argc < 8) return 2;
    const int64_t n_pad = std::atoll(argv[1]), m_pad = std::atoll(argv[2]);
    const int n_cu = std::atoi(argv[3]), n_cu_pipe = std::atoi(argv[4]), curve = std::atoi(argv[5]);
    const double noise = std::atof(argv[6]);
    std::mt19937_64 rng((unsigned long long)std::atoll(argv[7]));
    std::uniform_real_distribution<double> u(0.0, 1.0);
    ScheduleTable table;
    ScheduleEntry &e = schedule_entry(table, n_cu_pipe, n_pad, m_pad / 64, m_pad);
    int calls = 0, last_p = -99, last_g = -99;
    while (e.state != ScheduleEntry::SETTLED && calls < 400) {
        const ScheduleChoice ch = schedule_choose(e, true);
        double ms = step_ms(curve, e, ch.group, ch.pairs) * (1.0 + noise * u(rng));
        if (ch.pairs != last_p || ch.group != last_g) ms *= 1.05;        // the call after a change pays for it
        if (calls == 0) ms *= 3.0;                                       // cold
        // TUNER_SIM_WARM=w: a device whose clocks are still rising on the context's first calls -- call k is slower by
        // w exp(-k / 4) (a trace of a fresh context: 12 % on calls 2-3, 3 % ten calls later)
        if (const char *w = std::getenv("TUNER_SIM_WARM")) ms *= 1.0 + std::atof(w) * std::exp(-calls / 4.0);
        last_p = ch.pairs; last_g = ch.group;
        schedule_report(n_cu, n_cu_pipe, e, ch, 0, ms, ch.pairs < 0 ? 1700.0 : 0.0, ch.pairs < 0 ? 5100.0 : 0.0);
        ++calls;
    }
    if (argc > 8) {
        // a second shape next to the settled one (a grid of another size): it must start from the neighbour's split, not
        // from the plain sequence -- print what its FIRST call would run
        const int64_t m2 = std::atoll(argv[8]);
        ScheduleEntry &e2 = schedule_entry(table, n_cu_pipe, n_pad, m2 / 64, m2);
        const ScheduleChoice first = schedule_choose(e2, true);
        std::printf("%d %d %d\n", first.pairs, first.group, e2.all_pairs);
    }
    double best = 1e300;
    for (int grp = 0; grp <= 2; grp += 2)
        for (int p = -1; p <= e.all_pairs; ++p)
            if (schedule_is_candidate(e, grp, p)) { const double t = step_ms(curve, e, grp, p); if (t < best) best = t; }
    std::printf("%d %d %d %d %.4f %.4f\n", calls, e.cur, e.group, e.state == ScheduleEntry::SETTLED ? 1 : 0,
                step_ms(curve, e, e.group, e.cur), best);
    return 0;
}
