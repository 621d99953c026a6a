"""BfbTreeAllReducer on the GPU vs the oracle (bit for bit) and vs a plain sum
to the reference's tolerance (cedr_bfb_tree_allreduce.cpp:162-220)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ncell", [1, 3, 24, 1000, 1025, 5000])
@pytest.mark.parametrize("imbalanced", [False, True])
@pytest.mark.parametrize("transpose", [False, True])
def test_bfb_allreduce(gpu, oracle, ncell, imbalanced, transpose):
    ca = gpu
    nfield = 3
    rng = np.random.default_rng(ncell)
    send = rng.random(ncell * nfield)
    want = oracle.bfb_allreduce(oracle.Tree.make_1d(ncell, 1, False, imbalanced), ncell, 1, nfield,
                                transpose, [send])[0]
    b = ca.BfbTreeAllReducer(ca.Parallel.self_(), ca.Tree.make_1d(ncell, 1, False, imbalanced),
                             ncell, nfield)
    sd = ca.DeviceArray.from_numpy(send)
    rd = ca.DeviceArray(nfield)
    b.allreduce(sd, rd, transpose)
    ca.synchronize()
    got = rd.numpy()
    assert np.array_equal(got, want)
    plain = send.reshape(nfield, ncell).sum(1) if transpose else send.reshape(ncell, nfield).sum(0)
    re = np.abs(got - plain) / np.maximum(np.abs(got), np.abs(plain))
    assert re.max() <= max(2 * np.log(max(ncell, 2)) * np.finfo(float).eps, 0)
    # send and recv may alias
    b.allreduce(sd, sd, transpose)
    ca.synchronize()
    assert np.array_equal(sd.numpy()[:nfield], want)
