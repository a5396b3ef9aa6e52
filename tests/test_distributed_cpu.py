"""world_size-2 test of the multi-GPU path on CPU (gloo).

The product shards every sampler's lane range over ranks (interleaved 256-lane chunks, see
include/psdr_hip.h psdr_render_args.shard_rank/shard_count) and sums the partial [image | derivative]
buffers with ONE all_reduce.  The GPU kernels cannot run here, so each rank renders its shard with
the CPU oracle (test infrastructure) and then goes through the PRODUCT's rank discovery and
collective code (psdr_jit_amd._shard / _all_reduce); the reduced result must equal the unsharded frame.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import psdr_jit_amd as psdr
    import scenes
    from oracle import oracle as orc
    orc.set_num_threads(2)
    r, w = psdr._shard()
    assert (r, w) == (rank, world)
    spec = scenes.cbox_scene(24, 24, 16, 16, 16, param="light_x")
    sc = orc.OracleScene(spec, [0])
    img, dimg = sc.render_d(max_depth=2, seeds=(3, 4, 5), shard_rank=r, shard_count=w)
    buf = torch.from_numpy(np.stack([img, dimg]))
    psdr._all_reduce(buf, w > 1)
    # the product's renderD over ranks (psdr_jit_amd._render_terms) in both forms - the interior term's collective started ahead of the edge terms, and the
    # single collective of rounds 1-4 - with the oracle standing in for the kernels: the same frame
    def render(terms, continue_streams, image, derivative):
        i, d = sc.render_d(max_depth=2, seeds=(3, 4, 5), shard_rank=r, shard_count=w, terms=terms)
        image.copy_(torch.from_numpy(i)); derivative.copy_(torch.from_numpy(d))
    forms = {}
    for name, flag in (("split", "0"), ("single", "1")):
        os.environ["PSDR_SINGLE_COLLECTIVE"] = flag
        i, d = psdr._render_terms(render, 24 * 24, torch.device("cpu"), w, 7)
        forms[name] = np.stack([i.numpy(), d.numpy()])
    os.environ.pop("PSDR_SINGLE_COLLECTIVE", None)
    # PSDR_SHARD=rows (psdr_render_args.shard_mode 1): contiguous runs - rank r renders the pixel rows [r R, (r + 1) R) of the interior term, which is then ASSEMBLED by
    # all_gather_into_tensor (1 / world of the bytes per rank) while only the edge terms' derivative is summed.  24 rows over 2 ranks: 12 each; a 25-row frame tests the padding
    def render_rows(terms, continue_streams, image, derivative):
        i, d = sc.render_d(max_depth=2, seeds=(3, 4, 5), shard_rank=r, shard_count=w, shard_mode=1, terms=terms)
        image.copy_(torch.from_numpy(i)); derivative.copy_(torch.from_numpy(d))
    assert psdr._row_tile(24 * 24, 24, w, False) == 12 * 24
    i, d = psdr._render_terms(render_rows, 24 * 24, torch.device("cpu"), w, 7, tile=psdr._row_tile(24 * 24, 24, w, False))
    forms["rows"] = np.stack([i.numpy(), d.numpy()])
    i, d = psdr._render_terms(render_rows, 24 * 24, torch.device("cpu"), w, 7)          # (the same partition through one all_reduce of everything)
    forms["rows_reduced"] = np.stack([i.numpy(), d.numpy()])
    rows_part = sc.render_d(max_depth=2, seeds=(3, 4, 5), shard_rank=r, shard_count=w, shard_mode=1, terms=orc.TERM_INTERIOR)[0]
    assert np.abs(rows_part[(1 - r) * 12 * 24:(2 - r) * 12 * 24]).max() == 0.0 and np.abs(rows_part[r * 12 * 24:(r + 1) * 12 * 24]).max() > 0.0      # one block of rows per rank
    spec25 = scenes.cbox_scene(24, 25, 4, 4, 4, param="light_x")
    sc25 = orc.OracleScene(spec25, [0])
    def render_rows25(terms, continue_streams, image, derivative):
        i, d = sc25.render_d(max_depth=1, seeds=(3, 4, 5), shard_rank=r, shard_count=w, shard_mode=1, terms=terms)
        image.copy_(torch.from_numpy(i)); derivative.copy_(torch.from_numpy(d))
    i, d = psdr._render_terms(render_rows25, 24 * 25, torch.device("cpu"), w, 7, tile=psdr._row_tile(24 * 25, 24, w, False))
    forms["rows25"] = np.stack([i.numpy(), d.numpy()])
    if rank == 0:
        np.save(os.path.join(outdir, "full25.npy"), np.stack(sc25.render_d(max_depth=1, seeds=(3, 4, 5))))
        for k in ("rows", "rows_reduced", "rows25"):
            np.save(os.path.join(outdir, k + ".npy"), forms[k])
    if rank == 0:
        full = np.stack(sc.render_d(max_depth=2, seeds=(3, 4, 5)))
        np.save(os.path.join(outdir, "reduced.npy"), buf.numpy())
        np.save(os.path.join(outdir, "full.npy"), full)
        np.save(os.path.join(outdir, "part0.npy"), np.stack([img, dimg]))
        np.save(os.path.join(outdir, "split.npy"), forms["split"])
        np.save(os.path.join(outdir, "single.npy"), forms["single"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_reduce_to_full_frame(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    reduced, full, part0 = (np.load(os.path.join(tmp_path, f)) for f in ("reduced.npy", "full.npy", "part0.npy"))
    assert np.allclose(reduced[0], full[0], rtol=1e-5, atol=1e-7)
    assert np.allclose(reduced[1], full[1], rtol=1e-4, atol=1e-6)
    for form in ("split", "single"):
        got = np.load(os.path.join(tmp_path, form + ".npy"))
        assert np.allclose(got[0], full[0], rtol=1e-5, atol=1e-7) and np.allclose(got[1], full[1], rtol=1e-4, atol=1e-6), form
    # each rank really rendered only a part
    assert np.linalg.norm(part0[0]) < 0.9 * np.linalg.norm(full[0])
    # the contiguous partition (row tiles + all-gather), assembled and reduced, and on a frame whose rows do not divide
    for form, ref in (("rows", full), ("rows_reduced", full), ("rows25", np.load(os.path.join(tmp_path, "full25.npy")))):
        got = np.load(os.path.join(tmp_path, form + ".npy"))
        assert np.allclose(got[0], ref[0], rtol=1e-5, atol=1e-7) and np.allclose(got[1], ref[1], rtol=1e-4, atol=1e-6), form


def test_shard_is_identity_without_process_group():
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import __graft_entry__
    __graft_entry__.build()
    import psdr_jit_amd as psdr
    assert psdr._shard() == (0, 1)
