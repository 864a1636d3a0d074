// irv_plan.h -- the device-side state machine of the region-voting chain (k_voting.hip), shared with the CPU emulation
// (tests/emul/emul_irv.cpp): constants of the 16-bit state map, the control-block layout, the list layout and the pure
// function every block of kernel k evaluates to find out what this kernel has to do.
#pragma once
#include <stdint.h>
#include "adc_device_fn.h"

#define IRV_TILE 8
#define IRV_BIN_MASK 0x7FFu
#define IRV_FINAL 0x4000u
#define IRV_ELIG 0x8000u
#define IRV_PPT 4 // pixels per thread and block iteration of the BEGIN phase (one list-length atomic per 4 x blockDim pixels)

enum { IRV_NONE = 0, IRV_BEGIN, IRV_ROUND, IRV_FINAL_WB, IRV_DONE };
// ctrl layout (int32): state slot s at ctrl[16*s ..]: {did, pass, round, filled_any, n, rounds_total, evals, kdone};
// accumulator ring at ctrl[IRV_ACC + (k & 63)] (BEGIN: list length; ROUND: "a value changed")
#define IRV_ACC 64
#define IRV_CTRL_INTS 160
struct IrvState { int did, pass, round, filled_any, n, rounds, evals, kdone; }; // kdone: index of the kernel that found the chain finished
struct IrvPlan { int act; IrvState s; };

// One kernel type: kernel k reads the state its predecessor published (slot k & 1) and the predecessor's accumulator.
//   BEGIN    write the previous pass's fills back, mark the eligible pixels of the next list, build the work list
//   ROUND    round r of the pass: every open entry whose dependency box changed in round r-1 (round 0: every entry) votes.
//            Change tiles: kernel k stamps (k % 255) + 1 into plane k & 1 and reads the stamps of kernel k-1 in the other
//            plane -- both known at launch time, so the check can run before the state has arrived.
//   FINAL_WB write the last pass's fills back (and sum the per-wave evaluation counters into the state's evals)
ADC_HD IrvPlan irv_plan_from(IrvState s, int prev, int k) // s: the published state, prev: the predecessor's accumulator
{
    IrvPlan p;
    if (s.did == IRV_NONE) {
        p.act = IRV_BEGIN;
        s.pass = 0; s.round = 0; s.filled_any = 0; s.n = 0;
    } else if (s.did == IRV_BEGIN) {
        p.act = IRV_ROUND;
        s.n = prev; // list length
        s.round = 0;
    } else if (s.did == IRV_ROUND) {
        int fa = s.filled_any | ((s.round == 0 && prev != 0) ? 1 : 0); // a pass that fills anything does so in round 0
        if (prev != 0) { // the round changed something: next round
            p.act = IRV_ROUND;
            s.round++;
            s.filled_any = fa;
        } else { // a whole round without a change: the pass has converged
            bool fin = false;
            if (s.pass & 1) { // end of an iteration (multistep_refiner.cpp:167-171): nothing filled -> the rest are no-ops
                if (!fa) fin = true;
                fa = 0;
            }
            s.pass++;
            if (s.pass >= 10) fin = true;
            p.act = fin ? IRV_FINAL_WB : IRV_BEGIN;
            s.round = 0; s.filled_any = fa; s.n = 0;
        }
    } else {
        p.act = IRV_DONE;
    }
    if (p.act == IRV_ROUND) s.rounds++;
    if (p.act == IRV_DONE && s.did != IRV_DONE) s.kdone = k; // first kernel with nothing left to do
    s.did = p.act;
    p.s = s;
    return p;
}
ADC_HD IrvPlan irv_plan(const int32_t* ctrl, int k)
{
    const int32_t* in = ctrl + 16 * (k & 1);
    const IrvState s = {in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7]};
    return irv_plan_from(s, k > 0 ? ctrl[IRV_ACC + ((k - 1) & 63)] : 0, k);
}
ADC_HD void irv_publish(int32_t* ctrl, int k, const IrvState& s)
{
    int32_t* out = ctrl + 16 * ((k + 1) & 1);
    out[0] = s.did; out[1] = s.pass; out[2] = s.round; out[3] = s.filled_any; out[4] = s.n; out[5] = s.rounds; out[6] = s.evals; out[7] = s.kdone;
    ctrl[IRV_ACC + ((k + 2) & 63)] = 0;
}

// Work-list layout.  The chain's grid has G workgroups of WPB waves; a batch is B = 64 * WPB * G entries.  Entry i (in the
// order the BEGIN phase compacts them: raster order inside chunks of pixels) is evaluated by WORKGROUP (i % B) % G, where it
// sits in wave t % WPB, lane t / WPB with t = (i % B) / G: consecutive entries -- which tend to be dirty in the same
// rounds (a fill front is a few hundred adjacent pixels) -- land in different workgroups, so every workgroup's pool of
// dirty entries holds about dirty / G of them.  Stored so that the 64 entries of a wave are contiguous.
ADC_HD long irv_list_slot(long i, int G, int WPB)
{
    const long B = 64L * WPB * G, r = i % B;
    const long blk = r % G, t = r / G;
    return (i / B) * B + (blk * WPB + t % WPB) * 64 + t / WPB;
}
