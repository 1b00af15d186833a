"""N>1 path on CPU: world_size-2 gloo run of the sharded video flow (rank 0 prepares the
state, one broadcast of the blob, frames sharded, no per-frame collective).  The model
object here is the CPU oracle (the HIP Stylization has the same call surface); what is
under test is rerevst-code_amd/video.py + dist.py."""
import os
import sys
import importlib

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    torch.set_num_threads(2)
    pkg = importlib.import_module("rerevst-code_amd")
    V = importlib.import_module("rerevst-code_amd.video")
    D = importlib.import_module("rerevst-code_amd.dist")
    import rerevst_oracle as O
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    model = O.Stylization(pkg.synthetic_weights(0))
    frames = [pkg.synth_frame(i, 24, 32, kind="smooth") for i in range(5)]
    style = pkg.synth_style(32, 32, kind="smooth")
    out = V.stylize_video(model, frames, style, rank=r, world=w, broadcast=D.broadcast_state)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), state=model.get_state(),
             **{"f%d" % k: v for k, v in out.items()})
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_video_equals_single_process(tmp_path, pkg, oracle):
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    V = importlib.import_module("rerevst-code_amd.video")
    frames = [pkg.synth_frame(i, 24, 32, kind="smooth") for i in range(5)]
    style = pkg.synth_style(32, 32, kind="smooth")
    ref_model = oracle.Stylization(pkg.synthetic_weights(0))
    ref = V.stylize_video(ref_model, frames, style)
    r0, r1 = (np.load(tmp_path / ("rank%d.npz" % r)) for r in (0, 1))
    np.testing.assert_array_equal(r0["state"], r1["state"])          # broadcast delivered rank 0's blob
    np.testing.assert_array_equal(r0["state"], ref_model.get_state())
    got = {int(k[1:]): r[k] for r in (r0, r1) for k in r.files if k.startswith("f")}
    assert sorted(got) == [0, 1, 2, 3, 4]
    assert sorted(int(k[1:]) for k in r0.files if k.startswith("f")) == [0, 1]    # contiguous shards
    for i in range(5):
        assert got[i].shape == (24, 32, 3)
        np.testing.assert_allclose(got[i], ref[i], atol=2e-3)
