"""Per-stage parity harness: drives ONE HIP stage at a time through the C ABI's debug surface with
the ORACLE's outputs of the previous stage as inputs (so a defect in stage k cannot hide or fake a
defect in stage k+1), and compares bit-for-bit with the oracle's dump of that stage."""
import numpy as np

import adcensus_amd as A
from tests import cases


def diff(got, want):
    """(number of differing elements, first differing index or None)"""
    g = np.ascontiguousarray(got)
    w = np.ascontiguousarray(want)
    assert g.shape == w.shape and g.dtype == w.dtype, (g.shape, w.shape, g.dtype, w.dtype)
    if g.dtype.kind == "f":
        ne = g.view(np.uint32) != w.view(np.uint32)
    else:
        ne = g != w
    n = int(ne.sum())
    first = None
    if n:
        idx = tuple(int(i) for i in np.argwhere(ne)[0])
        first = {"idx": idx, "got": float(g[idx]) if g.dtype.kind == "f" else int(g[idx]),
                 "want": float(w[idx]) if w.dtype.kind == "f" else int(w[idx])}
    return n, first


def stage_report(left, right, opt, o, device=0, paper_modes=0):
    """o = oracle dump dict (all stages). Returns {stage: {"bad": n, "total": N, "first": ...}}.
    paper_modes: opt-in paper features set on the handle (then `o` must come from the port oracle run with the same modes)."""
    h, w = left.shape[:2]
    st = A.ADCensusStereo(device=device)
    popt = cases.to_product_option(opt)
    if not st.Initialize(w, h, popt):
        raise RuntimeError("Initialize failed: " + A.last_error())
    if paper_modes:
        st.set_paper_modes(paper_modes)
    rep = {}

    def rec(name, got, want):
        n, first = diff(got, want)
        rep[name] = {"bad": n, "total": int(np.asarray(want).size), "first": first}

    try:
        st.debug_set_images(left, right)
        st.debug_run(A.RUN_GRAY_CENSUS)
        rec("gray_left", st.debug_read(A.BUF_GRAY_LEFT), o["gray_left"])
        rec("gray_right", st.debug_read(A.BUF_GRAY_RIGHT), o["gray_right"])
        rec("census_left", st.debug_read(A.BUF_CENSUS_LEFT), o["census_left"])
        rec("census_right", st.debug_read(A.BUF_CENSUS_RIGHT), o["census_right"])

        st.debug_write(A.BUF_CENSUS_LEFT, o["census_left"])
        st.debug_write(A.BUF_CENSUS_RIGHT, o["census_right"])
        st.debug_run(A.RUN_COST)
        rec("cost_init", st.debug_read(A.BUF_VOLUME_A), o["cost_init"])

        st.debug_run(A.RUN_ARMS)
        rec("arms", st.debug_read(A.BUF_ARMS), o["arms"])
        rec("sup_count_h", st.debug_read(A.BUF_SUPCOUNT_H), o["sup_count_h"])
        rec("sup_count_v", st.debug_read(A.BUF_SUPCOUNT_V), o["sup_count_v"])

        st.debug_write(A.BUF_ARMS, o["arms"])
        st.debug_write(A.BUF_SUPCOUNT_H, o["sup_count_h"])
        st.debug_write(A.BUF_SUPCOUNT_V, o["sup_count_v"])
        st.debug_write(A.BUF_VOLUME_A, o["cost_init"])
        st.debug_run(A.RUN_AGGREGATE, 4)
        rec("cost_aggr", st.debug_read(A.BUF_VOLUME_A), o["cost_aggr"])
        # production variant: the first pass computes the matching cost itself from the images + census
        # (arg >= 100; ignores the cost volume) -- must give the same aggregated volume
        st.debug_write(A.BUF_VOLUME_A, np.zeros_like(o["cost_init"]))
        st.debug_run(A.RUN_AGGREGATE, 104)
        rec("cost_aggr(fused cost)", st.debug_read(A.BUF_VOLUME_A), o["cost_aggr"])
        # + ring depth chosen on the host from the maximum arms and same-direction pass pairs sharing a launch
        st.debug_run(A.RUN_ARMS)  # (recomputes armmax; the arms themselves equal the oracle's, checked above)
        st.debug_write(A.BUF_VOLUME_A, np.zeros_like(o["cost_init"]))
        st.debug_run(A.RUN_AGGREGATE, 304)
        rec("cost_aggr(fused cost + pass pairs)", st.debug_read(A.BUF_VOLUME_A), o["cost_aggr"])
        st.debug_write(A.BUF_VOLUME_A, o["cost_init"])
        st.debug_run(A.RUN_AGGREGATE, 204)
        rec("cost_aggr(pass pairs)", st.debug_read(A.BUF_VOLUME_A), o["cost_aggr"])

        st.debug_write(A.BUF_VOLUME_A, o["cost_aggr"])
        st.debug_run(A.RUN_SCANLINE, 4)
        rec("cost_so", st.debug_read(A.BUF_VOLUME_A), o["cost_so"])
        # the row passes were cut into verified segments when counter 5 > 1: no seam may have failed (counter 6) -- with the
        # production warm-up; a run with ADC_SO_WARM < 64 (test_scanline_segment_variants) expects failures and checks the redo
        import os
        rep["cost_so"]["segments"] = st.debug_counter(5)
        rep["cost_so"]["seam_fails"] = st.debug_counter(6) if st.debug_counter(5) > 1 else 0
        if int(os.environ.get("ADC_SO_WARM", "64")) >= 64:
            assert rep["cost_so"]["seam_fails"] == 0, rep["cost_so"]

        # production form of the scanline stage: the last pass also delivers the left-view winner-takes-all
        # (ADCensusStereo::ComputeDisparity, ADCensusStereo.cpp:188-243) -- isolated: oracle cost_aggr in, both
        # the optimised volume and the left disparity map compared
        st.debug_write(A.BUF_VOLUME_A, o["cost_aggr"])
        st.debug_write(A.BUF_DISP_LEFT, np.full((h, w), -7.0, np.float32))
        st.debug_run(A.RUN_SCANLINE, 104)
        rec("cost_so(fused wta)", st.debug_read(A.BUF_VOLUME_A), o["cost_so"])
        rec("disp_left_wta(fused in scanline)", st.debug_read(A.BUF_DISP_LEFT), o["disp_left_wta"])

        st.debug_write(A.BUF_VOLUME_A, o["cost_so"])
        st.debug_run(A.RUN_WTA)
        rec("disp_left_wta", st.debug_read(A.BUF_DISP_LEFT), o["disp_left_wta"])
        # (min_disparity > 0: the reference reads out of bounds in the last dmin columns -> cases.canonical)
        rec("disp_right_wta", cases.canonical("disp_right_wta", st.debug_read(A.BUF_DISP_RIGHT), opt),
            cases.canonical("disp_right_wta", o["disp_right_wta"], opt))

        if opt.do_lr_check:
            st.debug_write(A.BUF_DISP_LEFT, o["disp_left_wta"])
            st.debug_write(A.BUF_DISP_RIGHT, o["disp_right_wta"])
            st.debug_run(A.RUN_LRCHECK)
            rec("disp_after_lr", st.debug_read(A.BUF_DISP_LEFT), o["disp_after_lr"])
            rec("outlier_label", st.debug_read(A.BUF_OUTLIER_LABEL), o["outlier_label"])
            if opt.do_filling:
                st.debug_write(A.BUF_DISP_LEFT, o["disp_after_lr"])
                st.debug_write(A.BUF_OUTLIER_LABEL, o["outlier_label"])
                st.debug_run(A.RUN_REGION_VOTING)
                rec("disp_after_irv", st.debug_read(A.BUF_DISP_LEFT), o["disp_after_irv"])
                rep["disp_after_irv"]["voting_rounds_evals"] = st.voting_stats()
                # the same with a launch budget of four kernels: the chain stops early and is continued by the host
                before = st.debug_counter(1)
                st.debug_write(A.BUF_DISP_LEFT, o["disp_after_lr"])
                st.debug_run(A.RUN_REGION_VOTING, 4)
                rec("disp_after_irv(chain continued)", st.debug_read(A.BUF_DISP_LEFT), o["disp_after_irv"])
                if st.voting_stats()[0] > 2:
                    assert st.debug_counter(1) == before + 1, "the continuation path was not taken"
                st.debug_write(A.BUF_DISP_LEFT, o["disp_after_irv"])
                st.debug_run(A.RUN_INTERPOLATION)
                rec("disp_after_interp", st.debug_read(A.BUF_DISP_LEFT), o["disp_after_interp"])
        if opt.do_discontinuity_adjustment:
            st.debug_write(A.BUF_DISP_LEFT, o["disp_after_interp"])
            st.debug_write(A.BUF_VOLUME_A, o["cost_so"])
            st.debug_run(A.RUN_DISCONTINUITY)
            rec("disp_after_dda", st.debug_read(A.BUF_DISP_LEFT), o["disp_after_dda"])
        st.debug_write(A.BUF_DISP_LEFT, o["disp_after_dda"])
        st.debug_run(A.RUN_MEDIAN)
        rec("disp_final(median)", st.debug_read(A.BUF_DISP_LEFT), o["disp_final"])

        # whole pipeline through the drop-in entry point, twice (Match is stateless between calls)
        d1 = st.match(left, right)
        rec("match_final", d1, o["disp_final"])
        d2 = st.match(left, right)
        rec("match_repeat", d2, d1)
    finally:
        st.Release()
    return rep


def failing(rep):
    return {k: v for k, v in rep.items() if v["bad"]}
