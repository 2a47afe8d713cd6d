"""The engine-owned RCCL communicator (include/pdengine.h pd_comm_*) on the one GPU a test box has: a
world of one rank goes through librccl's init / all-gather / destroy for real; the N > 1 host logic is covered on the
CPU (tests/test_dist_cpu.py) and the N > 1 exchange itself needs more GPUs than a test may use."""
import numpy as np
import pytest
import torch

from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import weights as W
from prompt_diffusion_amd.dist import engine_all_gather_latents, engine_comm_init

pytestmark = pytest.mark.gpu


def test_gather_without_communicator_is_a_copy():
    eng = E.Engine(W.TINY, precision="f32")
    assert eng.comm_world() == (1, 0)
    x = np.random.default_rng(0).standard_normal((3, 4, 8, 8)).astype(np.float32)
    np.testing.assert_array_equal(eng.comm_all_gather(x), x)
    xd = torch.from_numpy(x).cuda()
    got = eng.comm_all_gather(xd)
    assert got.is_cuda and torch.equal(got, xd)
    eng.close()


def test_world_of_one_through_rccl(tmp_path):
    eng = E.Engine(W.TINY, precision="f32")
    engine_comm_init(eng, 0, 1, str(tmp_path / "id"))
    assert eng.comm_world() == (1, 0)
    with pytest.raises(E.PdError, match="already owns"):
        eng.comm_init(eng.comm_new_id(), 1, 0)
    x = np.random.default_rng(1).standard_normal((2, 4, 8, 8)).astype(np.float32)
    np.testing.assert_array_equal(engine_all_gather_latents(eng, x), x)            # host buffers: staged through the device
    xd = torch.from_numpy(x).cuda()
    assert torch.equal(engine_all_gather_latents(eng, xd, sizes=[2]), xd)           # device buffers: ncclAllGather in place
    eng.comm_destroy()
    assert eng.comm_world() == (1, 0)
    eng.comm_destroy()                                                               # idempotent
    eng.close()


def test_comm_argument_errors():
    eng = E.Engine(W.TINY, precision="f32")
    with pytest.raises(E.PdError, match="rank"):
        eng.comm_init(bytes(128), 2, 2)
    with pytest.raises(ValueError):
        eng.comm_init(bytes(5), 1, 0)
    eng.close()
