import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from bench import CHR20_LEN
from tophat_amd.synth import make_device_workload, make_scale_genome
seqs, genes = make_scale_genome(1, [CHR20_LEN], 20000, exon_len=300, intron_max=200000)
dev = torch.device('cuda', 0)
for n in (10_000_000, 40_000_000):
    w = make_device_workload(100, seqs, genes, None, n, dev, exon_len=300)
    for sd in ('left',):
        so = w[sd]['span_off']; cells = (so[1:] - so[:-1]).reshape(n, 4)
        allmapped = (cells > 0).all(1)
        seg = (w[sd]['seg_off'][1:] - w[sd]['seg_off'][:-1]).reshape(n, 4)
        print(n, sd, 'all span segs mapped', float(allmapped.float().mean()), 'missing>=1', float((~allmapped).float().mean()))
        # per 4M-chunk fraction of reads with all four segment hits but not contiguous (boundary reads): use hits' positions
        hits = w[sd]['hits'].reshape(-1, 4)
        so2 = w[sd]['seg_off']; cells2 = (so2[1:] - so2[:-1]).reshape(n, 4)
        ok = (cells2 == 1).all(1)
        idx = so2[:-1].reshape(n, 4)
        left0 = hits[idx[:, 0].clamp(max=hits.shape[0]-1).long(), 1]; left3 = hits[idx[:, 3].clamp(max=hits.shape[0]-1).long(), 1]
        span = (left3 - left0).abs()
        boundary = ok & (span != 75)
        print('   seg stage: all four segments mapped once', float(ok.float().mean()), 'and not contiguous', int(boundary.sum()))
        allmapped = ok
        for c in range(0, n, 4_000_000):
            print('   chunk', c // 1_000_000, 'M: boundary frac', float(boundary[c:c+4_000_000].float().mean()), 'missing frac', float((~allmapped[c:c+4_000_000]).float().mean()))
    del w; torch.cuda.empty_cache()
