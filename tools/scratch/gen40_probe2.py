import torch
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1)
N = 160_000_000
mapped = torch.rand(N, generator=g, device=dev) < 0.9
rows = torch.arange(N, device=dev, dtype=torch.int32)[:, None].expand(N, 4).contiguous()
cs = torch.cumsum(mapped.to(torch.int32), 0)
# chunked reference
parts = [mapped[i:i + (1 << 24)].sum().item() for i in range(0, N, 1 << 24)]
import itertools
acc = list(itertools.accumulate(parts))
bad_cs = [k for k, i in enumerate(range(0, N, 1 << 24)) if cs[min(i + (1 << 24), N) - 1].item() != acc[k]]
print('cumsum int32->', cs.dtype, 'bad chunks', bad_cs[:5], 'total', int(cs[-1]), acc[-1])
sel = rows[mapped]
ref = torch.cat([rows[i:i + (1 << 24)][mapped[i:i + (1 << 24)]] for i in range(0, N, 1 << 24)])
neq = (sel != ref).any(1)
print('rows[mapped] shape', tuple(sel.shape), 'differs from chunked in', int(neq.sum()), 'rows; first', int(neq.nonzero()[0]) if neq.any() else None)
nz = mapped.nonzero().squeeze(1)
sel2 = rows.index_select(0, nz)
print('index_select differs', int((sel2 != ref).any(1).sum()))
