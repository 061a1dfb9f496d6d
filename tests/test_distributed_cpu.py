"""world_size-2 gloo test of the shard-record exchange used on N GPUs: each rank rolls
out its own shard (oracle arithmetic on CPU), the records are all-gathered with the product's
`allgather_records`, and every rank must obtain the unsharded update."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "mppi-isaac_amd"))
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mppiisaac.planner.mppi import allgather_records, make_config
    from mppiisaac.utils.config_store import load_config
    from oracle.oracle import Oracle
    from scenes import panda_reach
    K, H = 128, 12
    scene, m, cfg_full, cost, dof, root = panda_reach(K=K, H=H)
    ex = load_config({"defaults": [{"mppi": "panda"}]})
    ex.mppi.num_samples, ex.mppi.horizon = K, H
    shard = make_config(ex.mppi, k_offset=rank * K // world, k_local=K // world, viz_link=scene.viz_link_index())
    o = Oracle("f64")
    U0 = np.zeros((H, 7))
    S, du, _ = o.rollout(m, shard, cost, dof, root, U0, o.sample(shard))
    records = torch.zeros((world, 2 + H * 7), dtype=torch.float64)
    records[rank] = torch.from_numpy(o.record(shard, S, du))
    allgather_records(records, rank)
    U, action, be = o.update(cfg_full, records.numpy(), U0)
    # the folded layout of the contact-scene kernels: `per` records per shard (here: the shard cut into 4 groups of samples),
    # written into this rank's rows and all-gathered in place - the combine of world * per records is the same update
    per, Kl = 4, K // world
    folded = torch.zeros((world * per, 2 + H * 7), dtype=torch.float64)
    for g in range(per):
        sub = make_config(ex.mppi, k_offset=rank * Kl + g * Kl // per, k_local=Kl // per, viz_link=scene.viz_link_index())
        sl = slice(g * Kl // per, (g + 1) * Kl // per)
        folded[rank * per + g] = torch.from_numpy(o.record(sub, S[sl], np.ascontiguousarray(du[:, :, sl])))
    allgather_records(folded, rank, per=per)
    Uf4, af4, bef4 = o.update(cfg_full, folded.numpy(), U0)
    assert np.allclose(af4, action, rtol=1e-10, atol=1e-12) and np.allclose(Uf4, U, rtol=1e-10, atol=1e-12)
    if rank == 0:  # unsharded reference
        Sf, duf, _ = o.rollout(m, cfg_full, cost, dof, root, U0, o.sample(cfg_full))
        Uf, af, bef = o.update(cfg_full, o.record(cfg_full, Sf, duf), U0)
        np.save(out + ".ref.npy", np.concatenate([af, bef]))
    np.save(f"{out}.{rank}.npy", np.concatenate([action, be]))
    dist.destroy_process_group()


def test_two_rank_record_exchange(tmp_path):
    out = str(tmp_path / "res")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    ref = np.load(out + ".ref.npy")
    r0, r1 = np.load(out + ".0.npy"), np.load(out + ".1.npy")
    np.testing.assert_array_equal(r0, r1)                       # every rank computes the same action
    np.testing.assert_allclose(r0, ref, rtol=1e-10, atol=1e-12)  # == single-shard result up to summation order
