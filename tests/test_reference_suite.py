"""Acceptance mode (A): the reference's OWN tests (tests/als_test.py, gpu_test.py, recommender_base_test.py of
benfred/implicit, unmodified) run against the reference's own Python model layer with its CUDA extension replaced by
this repository's shim (oracle/refsuite.py builds the tree in build/refsuite/ where /root/reference exists; it travels
to the GPU box with the snapshot).  This is the drop-in claim tested from the reference's side of the boundary."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITE = os.path.join(ROOT, "build", "refsuite")


def _run(args):
    # the reference's CPU extensions call BLAS from every OpenMP thread and OpenBLAS serves at most 128 concurrent callers:
    # on the 256-core GPU box its own calculate_loss (als_test.py::test_gpu_loss, CPU leg) segfaults with the default thread
    # count -- nothing of this repository is involved -- hence the caps
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([SUITE, os.path.join(SUITE, "tests"), ROOT]), OPENBLAS_NUM_THREADS="1",
               OMP_NUM_THREADS="32")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", *args], cwd=os.path.join(SUITE, "tests"),
                         env=env, capture_output=True, text=True, timeout=1500)
    tail = out.stdout[-3000:] + out.stderr[-1500:]
    m = re.search(r"(\d+) passed", out.stdout)
    return out.returncode, int(m.group(1)) if m else 0, tail


@pytest.mark.skipif(not os.path.isdir(SUITE), reason="build/refsuite not assembled (needs /root/reference at build time)")
def test_reference_gpu_tests_pass_over_the_shim(gpu):
    # one case is deselected: the CPU<->GPU conversion of BayesianPersonalizedRanking trains BPR on the GPU (bpr_update),
    # an algorithm outside this build's hot path; the ALS and LMF parametrisations of it run
    rc, passed, tail = _run(["gpu_test.py", "--deselect", "gpu_test.py::test_cpu_gpu_conversion[True-BayesianPersonalizedRanking]"])
    print(f"reference tests/gpu_test.py over the shim: {passed} passed (rc {rc})")
    assert rc == 0 and passed >= 44, tail


@pytest.mark.skipif(not os.path.isdir(SUITE), reason="build/refsuite not assembled (needs /root/reference at build time)")
def test_reference_als_tests_pass_over_the_shim(gpu):
    """als_test.py: the reference's GPUALSTest / GPUALSTestFloat16 mixin classes (recommend, similar_items, pickle,
    save/load, partial_fit, recalculate ...), the GPU/CPU loss parity test, to_cpu().to_gpu() -- plus its CPU cases."""
    rc, passed, tail = _run(["als_test.py"])
    print(f"reference tests/als_test.py over the shim: {passed} passed (rc {rc})")
    assert rc == 0 and passed >= 97, tail


_EVAL_SCRIPT = r"""
import os, sys, warnings, tempfile
import numpy as np
warnings.simplefilter("ignore")
import implicit.evaluation as ev                      # the reference's own evaluation.pyx (oracle/_ref build)
import implicit.cpu.als as ref_cpu                    # the reference's CPU model
import implicit_amd.gpu as gpu
from implicit_amd.gpu.als import AlternatingLeastSquares as HipALS
from implicit_amd.synthetic import synthetic_csr

C = synthetic_csr(3000, 1200, 150_000, seed=4)
train, test = ev.train_test_split(C, train_percentage=0.8, random_state=7)
train, test = train.tocsr().astype(np.float32), test.tocsr().astype(np.float32)
kw = dict(factors=64, regularization=0.05, iterations=8, random_state=11)
hip = HipALS(**kw)
hip.fit(train, show_progress=False)
ref = ref_cpu.AlternatingLeastSquares(**kw, num_threads=8)
ref.fit(train, show_progress=False)
m_hip = ev.ranking_metrics_at_k(hip, train, test, K=10, show_progress=False)   # drives hip.recommend() in batches
m_ref = ev.ranking_metrics_at_k(ref, train, test, K=10, show_progress=False)
print("metrics hip", m_hip)
print("metrics ref", m_ref)
for key in ("precision", "map", "ndcg", "auc"):
    assert abs(m_hip[key] - m_ref[key]) < 1e-3, (key, m_hip[key], m_ref[key])
fx = np.linalg.norm(hip.user_factors.to_numpy() - ref.user_factors) / np.linalg.norm(ref.user_factors)
print("factor distance after 8 free-running iterations", fx)
assert fx < 1e-3

# on-disk round trips in both directions (implicit/cpu/als.py:458-477, recommender_base.py:174-202)
d = tempfile.mkdtemp()
hip.save(os.path.join(d, "hip.npz"))
back = ref_cpu.AlternatingLeastSquares.load(os.path.join(d, "hip.npz"))      # stock implicit loads our file
np.testing.assert_array_equal(back.user_factors, hip.user_factors.to_numpy())
np.testing.assert_array_equal(back.item_factors, hip.item_factors.to_numpy())
assert back.factors == 64 and abs(back.regularization - 0.05) < 1e-12
ids_a, sc_a = back.recommend(5, train[5], N=10)
ids_b, sc_b = hip.recommend(5, train[5], N=10)
np.testing.assert_array_equal(ids_a, ids_b)
ref.save(os.path.join(d, "ref.npz"))
ours = HipALS.load(os.path.join(d, "ref.npz"))                                # and we load stock implicit's file
np.testing.assert_array_equal(ours.user_factors.to_numpy(), ref.user_factors)
np.testing.assert_array_equal(ours.item_factors.to_numpy(), ref.item_factors)
ids_c, _ = ours.recommend(5, train[5], N=10)
ids_d, _ = ref.recommend(5, train[5], N=10)
np.testing.assert_array_equal(ids_c, ids_d)
print("evaluation ok")
"""


@pytest.mark.skipif(not os.path.isdir(SUITE), reason="build/refsuite not assembled (needs /root/reference at build time)")
def test_ranking_metrics_and_cross_loading_with_stock_implicit(gpu):
    """the reference's `ranking_metrics_at_k` (implicit/evaluation.pyx:366-475) over a model trained HERE
    against the same metrics of the reference's CPU model trained from the same seed (p@10 / MAP / NDCG / AUC within
    1e-3), and model files crossing the boundary both ways: saved by implicit_amd.gpu.als -> loaded by stock
    implicit.cpu.als, and back."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([SUITE, ROOT]), OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="16")
    out = subprocess.run([sys.executable, "-c", _EVAL_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    print(out.stdout[-1500:])
    assert out.returncode == 0 and "evaluation ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
