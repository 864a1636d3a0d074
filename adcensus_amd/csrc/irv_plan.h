// irv_plan.h -- the device-side state machine of the region-voting chain (k_voting.hip), shared with the CPU emulation
// (tests/emul/emul_irv.cpp): constants of the 16-bit state map, the control-block layout and the pure function every
// block of kernel k evaluates to find out what this kernel has to do.
#pragma once
#include <stdint.h>
#include "adc_device_fn.h"

#define IRV_TILE 8
#define IRV_BIN_MASK 0x7FFu
#define IRV_FINAL 0x4000u
#define IRV_ELIG 0x8000u
#define IRV_PPT 8 // pixels per thread and block iteration of the BEGIN phase (one list-length atomic per 2048 pixels)

enum { IRV_NONE = 0, IRV_BEGIN, IRV_VOTE, IRV_CHECK, IRV_FINAL_WB, IRV_DONE };
// ctrl layout (int32): state slot s at ctrl[16*s ..]: {did, pass, round, filled_any, n, rounds_total, evals};
// accumulator ring at ctrl[IRV_ACC + (k & 63)]
#define IRV_ACC 64
struct IrvState { int did, pass, round, filled_any, n, rounds, evals, kdone; }; // kdone: index of the kernel that found the chain finished
struct IrvPlan { int act; IrvState s; int nwork; };

ADC_HD IrvPlan irv_plan(const int32_t* ctrl, int k)
{
    const int32_t* in = ctrl + 16 * (k & 1);
    IrvState s = {in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7]};
    const int prev = k > 0 ? ctrl[IRV_ACC + ((k - 1) & 63)] : 0;
    IrvPlan p;
    p.nwork = 0;
    if ((k & 1) == 0) { // kernel A
        if (s.did == IRV_NONE) {
            p.act = IRV_BEGIN;
            s.pass = 0; s.round = 0; s.filled_any = 0; s.n = 0;
        } else if (s.did == IRV_VOTE) {
            int fa = s.filled_any | ((s.round == 0 && prev != 0) ? 1 : 0); // a pass that fills anything does so in round 0
            if (prev != 0) { // the round changed something: next round
                p.act = IRV_CHECK;
                s.round++;
                s.filled_any = fa;
                p.nwork = s.n;
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
    } else { // kernel B
        if (s.did == IRV_BEGIN) {
            p.act = IRV_VOTE;
            s.n = prev; // list length
            p.nwork = prev;
        } else if (s.did == IRV_CHECK) {
            p.act = IRV_VOTE;
            p.nwork = prev; // dirty entries
        } else {
            p.act = IRV_DONE;
        }
        if (p.act == IRV_VOTE) { s.rounds++; s.evals += p.nwork; }
    }
    if (p.act == IRV_DONE && s.did != IRV_DONE) s.kdone = k; // first kernel with nothing left to do
    s.did = p.act;
    p.s = s;
    return p;
}
ADC_HD void irv_publish(int32_t* ctrl, int k, const IrvState& s)
{
    int32_t* out = ctrl + 16 * ((k + 1) & 1);
    out[0] = s.did; out[1] = s.pass; out[2] = s.round; out[3] = s.filled_any; out[4] = s.n; out[5] = s.rounds; out[6] = s.evals; out[7] = s.kdone;
    ctrl[IRV_ACC + ((k + 2) & 63)] = 0;
}

