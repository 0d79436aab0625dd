#!/usr/bin/env python3
"""BASELINE configs[3] on one GPU: both stages of the caller on a synthetic candidate set, text outputs included.

    stage 2  column encode + PileupModel forward + pileup.vcf rows                 (nanosnp_amd.predict.predict_pileup's kernels)
    stage 4  low-confidence candidates selected from the VCF                        (QUAL < 19, merge.select_groups' threshold)
    stage 5  read planes -> haplotype features -> HaplotypeModel forward -> haplotype.csv rows
    stage 6  merge of the two call sets                                             (nanosnp_amd.merge.merge_calls)

Synthetic: N windows from generator G2 for stage 2, generator G3 read planes for the selected sites (their generation is not
timed: the reference gets them from a BAM).  Reports wall time and sites/s per stage.  Not the bench metric."""
import sys, os, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib, host, merge
from nanosnp_amd.pileup_model import LSTMNetwork
from nanosnp_amd.predict import predict_haplotype, COV_CHANNELS
from nanosnp_amd.fixtures import load_pileup_weights, seeded_hap_weights

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
frac5 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15          # share of candidates sent to stage 5 at most
hap_prec = int(sys.argv[3]) if len(sys.argv) > 3 else 1
batch = 4096
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()

cols = host.synth_columns(77, N * 33, coverage=30.0, window=33)
d_bases = torch.from_numpy(cols.bases).to(dev); d_off = torch.from_numpy(cols.col_off).to(dev); d_ref = torch.from_numpy(cols.ref).to(dev)
model = LSTMNetwork().load_weight_list(load_pileup_weights())
ctx = model.ctx
fai = f"ctgS\t{N * 40 + 100}\t6\t60\t61\n"
names = ["ctgS"] * N
pos = (np.arange(N, dtype=np.int64) * 40 + 17)
table = host.ContigTable(names)

CH = 65536                                                       # windows per device pass (buffers are reused by torch's allocator)
cov_idx = torch.tensor(COV_CHANNELS, device=dev)
def stage2_gpu(n0, n1):
    c0, c1 = n0 * 33, n1 * 33
    b0, b1 = int(cols.col_off[c0]), int(cols.col_off[c1])
    off = d_off[c0:c1 + 1] - b0
    counts, depth, flags = ctx.pileup_encode_columns(d_bases[b0:b1], off, d_ref[c0:c1])
    centers = torch.arange(n1 - n0, dtype=torch.int64, device=dev) * 33 + 16
    gt, zy = ctx.pileup_forward_windows(counts, centers)
    ga, za, gm, zm, _ = ctx.pileup_postprocess(gt, zy)
    cov = counts.view(n1 - n0, 33, 18)[:, 16, :].index_select(1, cov_idx).to(torch.float32)
    return ga, za, gm, zm, cov
stage2_gpu(0, min(N, CH)); torch.cuda.synchronize()               # warm-up: code objects, workspaces, allocator
t0 = time.perf_counter()
# ---- stage 2 -----------------------------------------------------------------------------------------------------
parts = [stage2_gpu(n0, min(N, n0 + CH)) for n0 in range(0, N, CH)]
ga, za, gm, zm, cov = [torch.cat([p[i] for p in parts]) for i in range(5)]
refb = d_ref.view(N, 33)[:, 16].cpu().numpy()
refb = np.where(np.isin(refb, np.frombuffer(b"ACGT", np.uint8)), refb, ord("A")).astype(np.uint8)
torch.cuda.synchronize(); t_gpu2 = time.perf_counter() - t0
vcf_path = os.path.join(tmp, "pileup.vcf")
with open(vcf_path, "wb") as f:
    f.write(host.vcf_header(fai).encode())
    text, rows = host.vcf_format_batches(table, table.ids, pos, refb, ga.cpu().numpy(), za.cpu().numpy(), gm.cpu().numpy(),
                                         zm.cpu().numpy(), cov.cpu().numpy(), batch_size=1000)   # predict.py batch size
    f.write(text)
t2 = time.perf_counter() - t0
# ---- stage 4: low-confidence candidates ---------------------------------------------------------------------------
t0 = time.perf_counter()
vcf_text = open(vcf_path).read()
qual = {}
for line in vcf_text.splitlines():
    if line.startswith("#"): continue
    f_ = line.split("\t"); qual[int(f_[1])] = float(f_[5])
cand = np.array(sorted(p for p, q in qual.items() if q < 19.0), np.int64)
if cand.size > int(frac5 * N): cand = cand[:int(frac5 * N)]
t4 = time.perf_counter() - t0
n5 = int(cand.size)
# ---- stage 5 ------------------------------------------------------------------------------------------------------
pp = host.synth_hap_planes(100, max(n5, 1), 30, 90, 33); ph = host.synth_hap_planes(200, max(n5, 1), 30, 90, 11)
hctx = _lib.Context(0); hctx.hap_load_weights(seeded_hap_weights(12, H=256)); hctx.set_option("hap_precision", hap_prec)
cpos = [f"ctgS:{p}" for p in cand]
if len(sys.argv) > 4 and sys.argv[4] == "i8":                      # planes as a reader would produce them: int8
    pp = [a.astype(np.int8) for a in pp[:4]] + [pp[4]]; ph = [a.astype(np.int8) for a in ph[:4]] + [ph[4]]
predict_haplotype(hctx, [a[:256] for a in pp], [a[:256] for a in ph], cpos[:256], os.path.join(tmp, "warm.csv"), batch_size=4096)   # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
csv_path = os.path.join(tmp, "haplotype.csv")
predict_haplotype(hctx, [a[:n5] for a in pp], [a[:n5] for a in ph], cpos, csv_path, batch_size=4096)
torch.cuda.synchronize(); t5 = time.perf_counter() - t0
# ---- stage 6 ------------------------------------------------------------------------------------------------------
t0 = time.perf_counter()
merged = merge.merge_calls(vcf_text, open(csv_path).read())
t6 = time.perf_counter() - t0
print(f"stage 2: {N} candidate sites, GPU part {t_gpu2*1e3:.1f} ms ({N/t_gpu2/1e6:.1f} M sites/s), with VCF text {t2*1e3:.0f} ms ({N/t2/1e6:.2f} M sites/s), {rows} rows")
print(f"stage 4: {n5} low-confidence sites selected in {t4*1e3:.0f} ms (host text scan)")
print(f"stage 5: {n5} sites, features + HaplotypeModel forward (hap_precision {hap_prec}) + csv in {t5*1e3:.0f} ms ({n5/max(t5,1e-9)/1e3:.0f} k sites/s)")
print(f"stage 6: merge of {rows} + {n5} calls in {t6*1e3:.0f} ms -> {len(merged.splitlines())} lines")
print(f"total {t2 + t4 + t5 + t6:.2f} s for {N} candidates = {N/(t2+t4+t5+t6)/1e3:.0f} k candidates/s end to end on one GPU")
