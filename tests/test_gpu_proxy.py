"""SURVEY.md section 8(d) proxy at size, the GPU test (32 images; tests/proxy_agreement.py is the 256-image run whose result
is committed as profiles/r05/proxy_256.json): the whole network on the HIP kernels against the same model on the CPU
over the C oracle, 512 x 512, procedure of lib/detectors/ctdet.py:29-46."""
import pytest

pytestmark = pytest.mark.gpu


def test_detection_agreement_proxy_32_images():
    from tests import proxy_agreement as P
    r = P.compare(images=32, res=512, batch=8, seed=3)
    fp = r["fp32"]
    # (>= 0.999: one near-tie at the K-th place of one image may swap)
    assert fp["top100_agreement"] >= 0.999 and fp["site_agreement"] == 1.0 and fp["max_abs_diff_sigmoid_hm"] < 1e-4, fp
    # W4A8: 8-bit codes flip where fp32 re-association moves a value across a rounding boundary (tests/test_gpu_exact_codes.py
    # shows that this is the only cause) and ~70 re-quantising layers of a RANDOM-weight network amplify a flip -- frozen
    # ranges do not calm that.  The bar is the CPU path's own reproducibility at the same operating point (one thread
    # against many), and the order-independent site agreement.
    for mode in ("w4a8_frozen", "w4a8_running"):
        m, y = r[mode], r[mode]["cpu_vs_itself_one_thread"]
        assert m["site_agreement"] >= min(0.99, y["site_agreement"] - 0.01), (mode, m)
        assert m["top100_agreement"] >= y["top100_agreement"] - 0.15, (mode, m)
        for k in ("hm", "wh", "reg"):
            assert m["mean_abs_diff"][k] <= 2.5 * y["mean_abs_diff"][k] + 1e-3, (mode, k, m)
    b = r["w4a8_frozen_bytes"]
    assert b["site_agreement"] >= min(0.99, r["w4a8_frozen"]["cpu_vs_itself_one_thread"]["site_agreement"] - 0.01), b


def test_ap50_delta_proxy_32_images():
    """VERDICT r5 missing #1: an AP50 DELTA through the pinned VOC07 evaluator (tests/proxy_ap.py; 256 images:
    profiles/r06/proxy_ap.json).  |AP50(GPU W4A8) - AP50(CPU W4A8)| against pseudo ground truth from the fp32 CPU-oracle
    path stays within the CPU path's own one-thread-vs-many yardstick + 0.1 -- BASELINE.json's "AP50 within 0.1 of the
    reference" -- in every mode; fp32 on the GPU reproduces the ground truth it is scored against."""
    from tests import proxy_ap as A
    r = A.run(images=32, res=512, batch=8, seed=3, yard_images=32)
    # (a random-weight hm head ranks a few classes above the rest: the top-100 detections of an image -- and with them the
    # pseudo ground truth -- live in 2-3 of the 20 classes)
    assert r["ground_truth"]["boxes"] >= 200 and r["ground_truth"]["classes_with_boxes"] >= 2, r["ground_truth"]
    assert r["fp32"]["ap50_cpu"] == pytest.approx(1.0) and r["fp32"]["ap50_gpu"] >= 0.99, r["fp32"]
    for mode in ("w4a8_running", "w4a8_frozen", "w4a8_frozen_bytes"):
        m = r[mode]
        yard = abs(m["yardstick"]["delta_one_thread_minus_cpu"])
        assert abs(m["delta_gpu_minus_cpu"]) <= yard + 0.1, (mode, m)
        s = m["self_ground_truth"]["first_images"]
        assert s["ap50_gpu"] >= s["ap50_cpu_one_thread"] - 0.1, (mode, m)
