#!/usr/bin/env python3
"""Stress of the SHARDED runs: `world` processes in one gloo group share GPU 0 (as the 2- / 3-rank tests do) and work through random
inputs - mpileup text of 150 to 20,000 adversarial columns (so that ranks end up with no site, with fewer than the ten rows the
gt_output[ti] quirk reads, with batches that straddle two or three ranks), runs of window files of 0 to 3,000 windows - at batch sizes
1000 / 64 / 7 -, runs of haplotype site files of 0 to 400 sites; rank 0 compares what the group wrote (every rank formats its own rows, the text is gathered) with what it computes
alone (the same pipeline outside the group's sharding).  Test infrastructure.
    python tests/manual/sharded_stress.py [ROUNDS] [WORLD]"""
import os, socket, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def worker(rank, world, port, rounds, tmp, q):
    import numpy as np, torch
    import torch.distributed as dist
    import make_golden as mg
    from nanosnp_amd import _lib, host, sitefile
    from nanosnp_amd.fixtures import load_pileup_weights, seeded_hap_weights
    from nanosnp_amd.hap_pipeline import DeviceReference, predict_haplotype_bins
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import _format_rows, call_contig, predict_pileup_bins, stream_contig
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bad = 0
    try:
        m = LSTMNetwork().load_weight_list(load_pileup_weights())
        ctx = m.ctx
        hctx = _lib.Context(0); hctx.hap_load_weights(seeded_hap_weights(12, H=256))
        fai = "ctgA\t900000\t6\t60\t61\nctgB\t900000\t6\t60\t61\n"
        for r in range(rounds):
            rng = np.random.default_rng(5150 + r + int(os.environ.get("NSNP_STRESS_SEED", "0")))      # the same stream on every rank
            bs = int(rng.choice([1000, 64, 7]))
            # ---- text -> VCF rows ----
            n_cols = int(rng.choice([150, 900, 6000, 20000]))
            seq = rng.choice(np.frombuffer(b"ACGTacgtNn", np.uint8), n_cols + 3000, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02]).astype(np.uint8)
            pos = np.cumsum(np.where(rng.random(n_cols) < 0.01, rng.integers(2, 40, n_cols), 1)); pos = pos[pos < seq.size]
            cols = mg.adversarial_columns(rng, len(pos), seq[pos - 1])
            text = b"".join(b"chrA\t%d\t%c\t%d\t%s\t%s\n" % (p, seq[p - 1], len(c or "*"), (c or "*").encode("latin-1"), b"I" * max(1, len(c) // 2)) for p, c in zip(pos, cols))
            cb = int(rng.choice([1 << 40, 50_000, 9_000]))
            got = call_contig(m, text, "chrA", seq, batch_size=bs, chunk_bytes=cb)
            if rank == 0:
                rows = stream_contig(m, text, "chrA", seq, chunk_bytes=1 << 40)
                want = _format_rows(rows, "chrA", seq, bs, host.SCORE_FLOAT64) if rows.shape[0] else (b"", 0)
                ok1 = bytes(got[0]) == bytes(want[0]) and got[1] == rows.shape[0] and got[2] == want[1]
            else:
                ok1 = bytes(got[0]) == b"" and got[2] == 0
            # ---- window files -> VCF ----
            files = []
            for fi in range(int(rng.integers(1, 5))):
                n = int(rng.choice([0, 1, 9, 700, 3000]))
                path = os.path.join(tmp, f"r{r}_f{fi}.pd.bin")
                names = [str(c) for c in rng.choice(["ctgA", "ctgB"], n)]
                p_ = rng.integers(1, 800000, n).astype(np.int64)
                refb = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
                dt = str(rng.choice(["int16", "int32"]))
                if rank == 0:
                    cs = host.synth_columns(41000 + 10 * r + fi, max(n, 1) * 33, coverage=float(rng.choice([8, 30, 60])), window=33)
                    counts, _, _ = ctx.pileup_encode_columns(torch.from_numpy(cs.bases).cuda(), torch.from_numpy(cs.col_off).cuda(), torch.from_numpy(cs.ref).cuda())
                    x = ctx.pileup_gather_windows(counts, torch.arange(max(n, 1), dtype=torch.int64, device="cuda") * 33 + 16).cpu().numpy()[:n]
                    sitefile.write_pileup_bin(path, x, [f"{c}:{int(a)}:{'N' * 16}{chr(int(b))}{'N' * 16}" for c, a, b in zip(names, p_, refb)], matrix_dtype=dt)
                else:
                    rng.choice([8, 30, 60])                                                           # (keeps the streams in step)
                files.append(path)
            dist.barrier()
            ps = int(rng.choice([50, 777, 65536]))
            o = os.path.join(tmp, f"r{r}_sharded_{rank}.vcf")
            n_rows = predict_pileup_bins(m, files, fai, o, batch_size=bs, pass_sites=ps)
            ok2 = True
            if rank == 0:
                alone = os.path.join(tmp, f"r{r}_alone.vcf")
                n_alone = predict_pileup_bins(m, files, fai, alone, batch_size=bs, distributed=False)
                ok2 = open(o, "rb").read() == open(alone, "rb").read() and n_rows == n_alone
            else:
                ok2 = n_rows == 0 and not os.path.exists(o)
            # ---- haplotype site files -> csv (every rank formats the rows of its shard_range of every file) ----
            refs = {c: rng.choice(np.frombuffer(b"ACGTacgtN", np.uint8), int(rng.integers(300, 5000)), p=[.23, .23, .23, .23, .02, .02, .01, .01, .02]).astype(np.uint8)
                    for c in ("ctgA", "ctgB")}
            hfiles = []
            for fi in range(int(rng.integers(1, 4))):
                n = int(rng.choice([0, 1, 7, 130, 400]))
                Dp, Dh = int(rng.choice([1, 30, 90])), int(rng.choice([1, 25, 90]))
                contig = str(rng.choice(["ctgA", "ctgB", "ctgMissing"], p=[.5, .4, .1]))
                Lc = len(refs.get(contig, np.zeros(1000)))
                posn = np.sort(rng.choice(np.arange(-20, Lc + 40), n, replace=False)) if n else np.zeros(0, int)
                dt = str(rng.choice(["int8", "int32"]))
                path = os.path.join(tmp, f"r{r}_h{fi}.bin")
                if rank == 0:
                    pp = host.synth_hap_planes(100 * r + fi, n, 30, Dp, 33); ph = host.synth_hap_planes(100 * r + fi + 50, n, 30, Dh, 11)
                    planes = dict(zip(sitefile.HAP_PLANES, (ph[0], ph[3], ph[1], ph[2], pp[0], pp[3], pp[1], pp[2])))
                    sitefile.write_haplotype_bin(path, [f"{contig}:{p}" for p in posn], [[f"{contig}:{p + 37 * (k - 5)}" for k in range(11)] for p in posn],
                                                 planes, plane_dtype=dt)
                hfiles.append(path)
            dist.barrier()
            href = DeviceReference(refs, 0)
            ho = os.path.join(tmp, f"r{r}_hap_{rank}.csv")
            hps = int(rng.choice([50, 128, 16384]))
            h_rows = predict_haplotype_bins(hctx, hfiles, href, ho, pass_sites=hps)
            if rank == 0:
                h_alone = os.path.join(tmp, f"r{r}_hap_alone.csv")
                n_h = predict_haplotype_bins(hctx, hfiles, DeviceReference(refs, 0), h_alone, distributed=False)
                ok3 = open(ho, "rb").read() == open(h_alone, "rb").read() and h_rows == n_h
            else:
                ok3 = h_rows == 0
            bad += not (ok1 and ok2 and ok3)
            if rank == 0:
                print(f"round {r}: batch size {bs}; text of {len(pos)} columns, {got[1]} sites in chunks of {cb}: {'identical' if ok1 else 'DIFFERS'}; "
                      f"{len(files)} window files in passes of {ps}, {n_rows} rows: {'identical' if ok2 else 'DIFFER'}; "
                      f"{len(hfiles)} haplotype files in passes of {hps}, {h_rows} rows: {'identical' if ok3 else 'DIFFER'}", flush=True)
            dist.barrier()
    finally:
        q.put((rank, bad))
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=worker, args=(r, world, port, rounds, tmp, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=1500) for _ in procs]
        for p in procs:
            p.join(60)
    bad = sum(b for _, b in res) + sum(p.exitcode != 0 for p in procs)
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
