"""GPU: the HIP path through the C ABI reproduces the committed fixtures (both stages, device-to-device sets)."""
import copy

import pytest

import orc
from golden_util import CASES, events_text, fusions_text, load
from tophat_amd import host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_hip_path_reproduces_fixture(name, tmp_path):
    c = load(name)
    seqs = [None if s is None else orc.fold_genome_char(s) for s in c["seqs"]]
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        runs, base = [], 0
        for side, b in c["seg_batches"]:
            p = copy.copy(c["p"])
            p.read_side = side
            runs.append((p, ctx.upload_batch(b, ordinal_base=base)))
            base += b.n_reads
        ev = ctx.segjuncs(runs)
        exp = dict(c["exp"])
        exp_fus = exp.pop("fusions", None)
        assert events_text(ev, c["names"], tmp_path) == exp
        if c["fusion"]:
            fus = ctx.fusions(runs, c["fusion_ignore"])
            assert fusions_text(fus, ev.juncs, c["names"], tmp_path) == exp_fus
        ctx.span_sets_from_segjuncs()
        for sd, sb in c["span_batches"].items():
            got = ctx.spanning(c["p"], [ctx.upload_span_batch(sb)])
            assert [a.sam_fields(int(sb.read_id[a.read_idx]), c["names"]) for a in got] == c["exp_span"][sd]
