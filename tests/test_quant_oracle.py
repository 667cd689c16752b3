"""Pins oracle/quant.py (the restatement of the reference's quantisation arithmetic and stage
compositions) against golden vectors produced by the REFERENCE'S OWN Python modules
(tests/golden/make_golden.py).  quant_ref.npz involves no oracle code at all.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import quant as Q

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, name)).items()}


def test_quantact_state_and_codes_exact():
    z = load("quant_ref.npz")
    st = Q.QuantActState(bits=8)
    for it in range(4):
        y, q = st(z["act_x%d" % it], running=True, return_codes=True)
        assert torch.equal(st.x_min, z["act_min%d" % it]) and torch.equal(st.x_max, z["act_max%d" % it])
        assert torch.equal(q, z["act_q%d" % it])
        assert torch.equal(y, z["act_y%d" % it])
    # codes really leave the int8 range (unclamped asymmetric branch)
    assert z["act_q1"].max() > 127 or z["act_q1"].min() < -128


def test_quantact_percentile_exact():
    z = load("quant_ref.npz")
    st = Q.QuantActState(bits=8)
    for it in range(2):
        y = st(z["pact_x%d" % it], running=True, percentile=True)
        assert torch.equal(st.x_min, z["pact_min%d" % it]) and torch.equal(st.x_max, z["pact_max%d" % it])
        assert torch.equal(y, z["pact_y%d" % it])


@pytest.mark.parametrize("tag,pct", [("n", False), ("p", True)])
def test_weight_quant_exact(tag, pct):
    z = load("quant_ref.npz")
    wq = Q.weight_fake_quant(z["qconv_%s_w" % tag], 4, pct)
    y = torch.nn.functional.conv2d(z["qconv_%s_x" % tag], wq, z["qconv_%s_b" % tag])
    # conv outputs: summation order depends on the CPU thread count, so compare to 1e-5 -- a single
    # wrong 4-bit weight code would move y by >= |w|/7 * |x| ~ 1e-2
    assert (y - z["qconv_%s_y" % tag]).abs().max().item() < 1e-5
    wf, bf = Q.fold_bn(z["qbn_%s_w" % tag], None, z["qbn_%s_bn_weight" % tag], z["qbn_%s_bn_bias" % tag],
                       z["qbn_%s_bn_running_mean" % tag], z["qbn_%s_bn_running_var" % tag], 1e-5)
    y2 = torch.nn.functional.conv2d(z["qbn_%s_x" % tag], Q.weight_fake_quant(wf, 4, pct), bf)
    assert (y2 - z["qbn_%s_y" % tag]).abs().max().item() < 1e-4
    assert torch.equal(Q.weight_fake_quant(z["qdw_%s_w" % tag], 4, pct), z["qdw_%s_wq" % tag])


def test_stage_fp32_matches_reference_module():
    z = load("stage_fp32.npz")
    r = Q.stage_fp32(z["x"], z["w_scale"], z["b_scale"], z["w_dw"], z["w_pw"])
    assert (r["s"] - z["s"]).abs().max().item() < 5e-6     # conv summation order (threads)
    assert (r["y"] - z["y"]).abs().max().item() < 1e-4
    assert z["s"].min().item() == -7.0 and z["s"].max().item() == 8.0   # both clamps exercised


@pytest.mark.parametrize("tag,pct", [("n", False), ("p", True)])
def test_stage_w4a8_matches_reference_module(tag, pct):
    z = load("stage_w4a8.npz")
    bn = (z[tag + "_bn_weight"], z[tag + "_bn_bias"], z[tag + "_bn_running_mean"],
          z[tag + "_bn_running_var"], 1e-5)
    act_s, act_d, act_r = Q.QuantActState(), Q.QuantActState(), Q.QuantActState()
    for it in range(3):
        r = Q.stage_w4a8(z["%s_x%d" % (tag, it)], z[tag + "_w_scale"], z[tag + "_b_scale"],
                         z[tag + "_w_dw"], z[tag + "_w_pw"], bn, act_s, act_d, wt_percentile=pct)
        # (conv summation order may differ by thread count: ranges to 1e-6 rel, values to 1 LSB)
        def close(a, b, tol):
            return (a - b).abs().max().item() <= tol
        assert close(act_s.x_min, z["%s_smin%d" % (tag, it)], 5e-6)
        assert close(act_s.x_max, z["%s_smax%d" % (tag, it)], 5e-6)
        assert close(r["s"], z["%s_s%d" % (tag, it)], 5e-6)
        assert close(r["d"], z["%s_d%d" % (tag, it)], 1e-5)
        assert close(act_d.x_min, z["%s_dmin%d" % (tag, it)], 1e-5)
        assert close(r["d_q"], z["%s_dq%d" % (tag, it)], 1e-5)
        assert close(r["y"], z["%s_y%d" % (tag, it)], 1e-4)
        post = act_r(torch.relu(r["y"]))
        assert close(post, z["%s_r%d" % (tag, it)], 1e-4)
        assert close(act_r.x_max, z["%s_rmax%d" % (tag, it)], 1e-4)


def test_deform_raw_regression():
    """oracle-only vectors: guards the C oracle against accidental change."""
    from oracle import dcn as O
    z = load("deform_raw.npz")
    for tag in ("a", "b"):
        N, C, H, W, Co, k, s, p, d, Gr, DG = [int(v) for v in z[tag + "_cfg"]]
        cfg = (s, p, d, Gr, DG)
        y = O.deform_conv_forward(z[tag + "_x"], z[tag + "_off"], z[tag + "_w"], *cfg)
        assert torch.equal(y, z[tag + "_y"])
        assert y[0, :, 0, 0].abs().max().item() == 0.0      # out-of-range samples


def _head_params(z):
    bn1 = (z["bn1_weight"], z["bn1_bias"], z["bn1_mean"], z["bn1_var"])
    bn2 = (z["bn2_weight"], z["bn2_bias"], z["bn2_mean"], z["bn2_var"])
    return z["w1"], bn1, z["w2"], bn2, z["w3"], z["b3"]


def test_head_fp32_matches_reference_sequential():
    z = load("head_w4a8.npz")
    r = Q.head_fp32(z["x0"], *_head_params(z))
    assert (r["out"] - z["fp32_out"]).abs().max().item() < 1e-5


@pytest.mark.parametrize("tag,pct", [("n", False), ("p", True)])
def test_head_w4a8_matches_reference_module(tag, pct):
    """oracle/quant.py head_w4a8 vs the reference's QuantDepthwiseNode over 3 forwards (EMA state)."""
    torch.set_num_threads(1)
    z = load("head_w4a8.npz")
    a1, a3 = Q.QuantActState(bits=8), Q.QuantActState(bits=8)
    for it in range(3):
        r = Q.head_w4a8(z["x%d" % it], *_head_params(z), a1, a3, wt_percentile=pct)
        assert (a1.x_min - z["%s_a1min%d" % (tag, it)]).abs().item() < 1e-5
        assert (a1.x_max - z["%s_a1max%d" % (tag, it)]).abs().item() < 1e-5
        assert (a3.x_min - z["%s_a3min%d" % (tag, it)]).abs().item() < 1e-5
        assert (a3.x_max - z["%s_a3max%d" % (tag, it)]).abs().item() < 1e-5
        lsb1 = (a1.x_max - a1.x_min).item() / 255.0
        lsb3 = (a3.x_max - a3.x_min).item() / 255.0
        d1 = (r["y1q"] - z["%s_y1q%d" % (tag, it)]).abs()
        d3 = (r["y2q"] - z["%s_y2q%d" % (tag, it)]).abs()
        # same ops, same thread count: identical up to a rare one-LSB code flip
        assert d1.max().item() <= 1.01 * lsb1 and (d1 > 1e-6).float().mean().item() < 1e-3
        assert d3.max().item() <= 1.01 * lsb3 and (d3 > 1e-6).float().mean().item() < 5e-3
        assert (r["out"] - z["%s_out%d" % (tag, it)]).abs().max().item() < 0.05


def _unit_params(z, u, stride):
    p = {}
    for k in ("1", "2", "3") + (("4", "5") if stride == 2 else ()):
        p["w" + k] = z["u%d_w%s" % (u, k)]
        p["bn" + k] = tuple(z["u%d_bn%s" % (u, k)][i] for i in range(4))
    return p


def test_base_node_oracle_matches_reference_units():
    """oracle/quant.py base_node_w4a8 vs two chained reference QuantBaseNode units (stride 2, stride 1)
    sharing one block-output QuantAct, 3 forwards (tests/golden/base_nodes.npz)."""
    torch.set_num_threads(1)
    z = load("base_nodes.npz")
    shared = Q.QuantActState(bits=8)
    acts = [{k: Q.QuantActState(bits=8) for k in ("act1", "act2", "act4")},
            {k: Q.QuantActState(bits=8) for k in ("act1", "act2")}]
    for it in range(3):
        y0 = Q.base_node_w4a8(z["x%d" % it], _unit_params(z, 0, 2), acts[0], shared, 2)
        y1 = Q.base_node_w4a8(y0, _unit_params(z, 1, 1), acts[1], shared, 1)
        assert (shared.x_min - z["shared_min%d" % it]).abs().item() < 1e-5
        assert (shared.x_max - z["shared_max%d" % it]).abs().item() < 1e-5
        for u, names in ((0, ("act1", "act2", "act4")), (1, ("act1", "act2"))):
            for k in names:
                assert (acts[u][k].x_max - z["u%d_quant_%s_max%d" % (u, k, it)]).abs().item() < 1e-5
                assert (acts[u][k].x_min - z["u%d_quant_%s_min%d" % (u, k, it)]).abs().item() < 1e-5
        lsb = (shared.x_max - shared.x_min).item() / 255.0
        for got, ref in ((y0, z["y0_%d" % it]), (y1, z["y1_%d" % it])):
            d = (got - ref).abs()
            assert d.max().item() <= 1.01 * lsb and (d > 1e-6).float().mean().item() < 5e-3
