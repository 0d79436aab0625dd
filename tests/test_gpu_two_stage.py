"""BASELINE configs[3] in miniature on the GPU: stage 2 (mpileup text -> encode -> PileupModel -> pileup.vcf), stage 4 (group
selection), stage 5 (read planes -> haplotype features -> HaplotypeModel -> haplotype.csv) and stage 6 (merge) on the encode_g1
fixture, every stage compared with what the reference's own code produced for it (tests/golden/two_stage.npz;
run_caller.sh:109-141).  Probabilities differ from CPU torch by ~1e-7, which can move a QUAL by one unit of its second decimal."""
import gzip

import numpy as np
import pytest

from nanosnp_amd import host, merge
from tests.helpers import golden, seeded_hap_weights
from tests.test_two_stage_host import TWO_STAGE_HAP_WEIGHTS

pytestmark = pytest.mark.gpu


def _same_up_to_qual(got_lines, want_lines, qual_col, gq_in_sample=True, explain=None):
    """rows equal except QUAL (and the GQ copy of it).  explain(got fields, reference QUAL) must hold for EVERY differing row: the
    reference's QUAL is the QUAL of a probability at most 1e-6 away from the GPU's own (tests/helpers.py qual_reachable) - there is no
    budget of tolerated rows.  Returns the positions (field 1) of the rows that differ."""
    assert len(got_lines) == len(want_lines)
    moved = []
    for g, w in zip(got_lines, want_lines):
        if g == w:
            continue
        gf, wf = g.split("\t"), w.split("\t")
        assert len(gf) == len(wf)
        for k, (a, b) in enumerate(zip(gf, wf)):
            if k == qual_col:
                assert abs(float(a) - float(b)) <= 0.0101, (g, w)
            elif gq_in_sample and k == len(gf) - 1:
                sa, sb = a.split(":"), b.split(":")
                assert sa[0] == sb[0] and sa[2:] == sb[2:] and abs(int(sa[1]) - int(sb[1])) <= 1, (g, w)
            else:
                assert a == b, (g, w)
        if explain is not None:
            assert explain(gf, float(wf[qual_col])), (g, w)
        moved.append(gf[1])
    return moved


def test_two_stage_chain_matches_the_reference_stage_by_stage(tmp_path, pileup_weights):
    from nanosnp_amd import _lib
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_variants
    from nanosnp_amd.predict import predict_haplotype
    z = np.load(golden("two_stage.npz"))
    fa = tmp_path / "ref.fa"
    fa.write_bytes(gzip.open(golden("encode_g1.fa.gz")).read())
    mp = tmp_path / "chrS.mpileup"
    mp.write_bytes(gzip.open(golden("encode_g1.mpileup.gz")).read())
    # ---- stage 2 ----
    m = LSTMNetwork().load_weight_list(pileup_weights)
    vcf_path = tmp_path / "pileup.vcf"
    call_variants(m, [("chrS", str(mp))], str(fa), "chrS\t6100\t6\t60\t61\n", str(vcf_path))
    vcf = vcf_path.read_text()
    want_vcf = bytes(z["vcf_s2"]).decode()
    # the GPU's own probabilities of the called sites (the same kernels, the same windows): a differing QUAL must be reachable from them
    import torch
    from tests.helpers import qual_reachable
    text = mp.read_bytes()
    pos_all, col_off, bases = host.mpileup_parse(text)
    seq0 = host.fasta_load_contig(str(fa), "chrS")
    dev = torch.device("cuda", m.ctx.device)
    counts, _, flags = m.ctx.pileup_encode_columns(torch.from_numpy(bases).to(dev), torch.from_numpy(col_off).to(dev), torch.from_numpy(seq0[pos_all - 1]).to(dev))
    centers, _ = m.ctx.pileup_select_sites(torch.from_numpy(pos_all).to(dev), flags)
    gt_p, zy_p = m.ctx.pileup_forward_windows(counts, centers)
    p_of = {int(pos_all[c]): (float(g.max()), float(zz.max())) for c, g, zz in zip(centers.cpu().numpy(), gt_p.cpu().numpy(), zy_p.cpu().numpy())}
    moved_vcf = _same_up_to_qual(vcf.splitlines(), want_vcf.splitlines(), 5,
                                 explain=lambda gf, q_ref: qual_reachable(q_ref, p_of[int(gf[1])][1], p_of[int(gf[1])][0], refcall=gf[6] == "RefCall"))
    print("stage-2 rows whose QUAL differs by a rounding boundary:", len(moved_vcf))
    # ---- stage 4: candidates below QUAL 19 with five confident heterozygous neighbours each side ----
    groups = merge.select_groups(vcf, quality_threshold=19.0, adjacent_size=5, support_quality=14.0)["chrS"]
    gpos = np.array([[p for p, _, _ in g] for g in groups], np.int64)
    assert np.array_equal(gpos, z["group_pos"])
    # ---- stage 5 ----
    seq = host.fasta_load_contig(str(fa), "chrS")
    cands = [f"chrS:{p}" for p in gpos[:, 5]]
    rp = host.haplotype_ref_rows({"chrS": seq}, cands, 33)
    rh = host.haplotype_ref_rows({"chrS": seq}, cands, 11, position_lists=[[f"chrS:{p}" for p in row] for row in gpos])
    pp = [z[f"p_{n}"] for n in ("seq", "bq", "mq", "hap")] + [rp]            # int8 planes, as a reader would hand them over
    ph = [z[f"h_{n}"] for n in ("seq", "bq", "mq", "hap")] + [rh]
    hctx = _lib.Context(0)
    hctx.hap_load_weights(seeded_hap_weights(**TWO_STAGE_HAP_WEIGHTS))
    csv_path = tmp_path / "haplotype.csv"
    predict_haplotype(hctx, pp, ph, cands, str(csv_path), batch_size=7)
    csv = csv_path.read_text()
    # stage 5 likewise: QUAL of haplotype.csv = calculate_score(max genotype probability) (predict_dev.py:40-47)
    xs = [hctx.hap_features(*[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in pl[:4]], torch.from_numpy(pl[4]).to(dev)) for pl in (pp, ph)]
    hg, _ = hctx.hap_forward(xs[0], xs[1])
    hp = {int(c.split(":")[1]): float(v) for c, v in zip(cands, hg.max(1).values.cpu().numpy())}
    moved_csv = _same_up_to_qual(csv.splitlines(), bytes(z["csv"]).decode().splitlines(), 3, gq_in_sample=False,
                                 explain=lambda gf, q_ref: qual_reachable(q_ref, hp[int(gf[1])]))
    hctx.close()
    # ---- stage 6 ----
    for q in (15.0, 19.0):
        got = merge.merge_calls(vcf, csv, q).splitlines()
        want = bytes(z[f"merged_q{int(q)}"]).decode().splitlines()
        # a merged row can only differ where the stage-2 or stage-5 row it was made from differed
        assert set(_same_up_to_qual(got, want, 5)) <= set(moved_vcf) | set(moved_csv)
    assert sum("\tH\t" in l for l in merge.merge_calls(vcf, csv, 19.0).splitlines()) == 9


def _sharded_worker(rank, world, port, tmp, q):
    """one of `world` processes of the s1 + s2 pipeline (all on cuda:0 of the one-GPU box; the result gather over gloo)"""
    import os
    import torch.distributed as dist
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_variants
    from tests.helpers import load_pileup_weights
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = LSTMNetwork().load_weight_list(load_pileup_weights())
        out = os.path.join(tmp, f"sharded_{rank}.vcf")
        rows = call_variants(m, [("chrS", os.path.join(tmp, "chrS.mpileup"))], os.path.join(tmp, "ref.fa"), "chrS\t6100\t6\t60\t61\n", out)
        q.put((rank, rows, os.path.exists(out)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_column_sharded_pipeline_writes_the_single_process_vcf(tmp_path, pileup_weights, world):
    """ranks encode their column range (+ 16-column halo), call the sites centred in it and gather the calls to rank 0:
    the VCF is byte-identical to the single-process one (nanosnp_amd.pipeline.call_contig under torch.distributed)"""
    import socket
    import torch.multiprocessing as mp
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_variants
    (tmp_path / "ref.fa").write_bytes(gzip.open(golden("encode_g1.fa.gz")).read())
    (tmp_path / "chrS.mpileup").write_bytes(gzip.open(golden("encode_g1.mpileup.gz")).read())
    m = LSTMNetwork().load_weight_list(pileup_weights)
    single = tmp_path / "single.vcf"
    rows1 = call_variants(m, [("chrS", str(tmp_path / "chrS.mpileup"))], str(tmp_path / "ref.fa"), "chrS\t6100\t6\t60\t61\n", str(single))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert outs[0] == (0, rows1, True) and all(o[1] == 0 and not o[2] for o in outs[1:])
    assert (tmp_path / "sharded_0.vcf").read_bytes() == single.read_bytes()
