"""world_size-2 test of the multi-GPU host logic on CPU (gloo): utterance sharding + output
gather reproduce the single-process result.  The compute function here is the ORACLE (the
product path has no CPU implementation); what is under test is fnssl/dist.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, rs_randn
from fnssl import dist as fdist


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 5, 32, 33):
        for w in (1, 2, 3, 8):
            spans = [fdist.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        fdist.shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nb, out_dir):
    for p in (os.path.join(ROOT, "fn-ssl_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from fnssl import dist as fd
    from fnssl import weights as W
    from oracle import fnssl_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sd = W.make_fnssl_state(11)
    batch = torch.from_numpy(rs_randn(12, (nb, 2, 512 + 11 * 256), 0.05))

    def predict(shard):
        return torch.from_numpy(O.predict_step(sd, shard.numpy(), "MM", True))

    got = fd.predict_sharded(predict, batch)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), got.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("nb", [3, 1])
def test_two_rank_sharded_predict_matches_single_process(tmp_path, nb):
    from fnssl import weights as W
    from oracle import fnssl_oracle as O
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, nb, str(tmp_path)), nprocs=world, join=True)
    sd = W.make_fnssl_state(11)
    batch = rs_randn(12, (nb, 2, 512 + 11 * 256), 0.05)
    want = O.predict_step(sd, batch, "MM", True)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        np.testing.assert_array_equal(got, want)     # uneven shards (2+1, 1+0) included


# ---- training: the gradient exchange step (fnssl.train.sync_gradients) over gloo ----------------------
def _grad_worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "fn-ssl_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from fnssl import train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = train.shard_utterances(5, rank, world)
    # the "gradient" of a rank = sum over its utterances of a per-utterance vector
    per_utt = torch.from_numpy(rs_randn(70, (5, 1000)))
    flat = per_utt[lo:hi].sum(dim=0)
    scale = train.sync_gradients(flat)
    np.save(os.path.join(out_dir, "g%d.npy" % rank), (flat * scale).numpy())
    dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo(tmp_path):
    from fnssl import train
    assert [train.shard_utterances(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    spans = [train.shard_utterances(5, r, 2) for r in range(2)]
    assert spans == [(0, 3), (3, 5)]
    assert train.sync_gradients(torch.ones(3)) == 1.0          # no process group: identity
    port = _free_port()
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = rs_randn(70, (5, 1000)).sum(axis=0) / 2.0
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "g%d.npy" % r))
        assert np.allclose(got, want, rtol=1e-6, atol=1e-6)
