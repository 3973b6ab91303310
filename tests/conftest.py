import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'e2e-asr-pytorch_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, so `-m "not gpu"` and a bare
    `pytest tests/` both work in the CPU-only container."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _no_silent_abort(request):
    """A persistent launch that gave up inside a GPU test sets the sticky status word (src/hipabi.py) and the optimizer
    then refuses every later update of the process: fail the test that caused it, not a later one.  Tests that provoke an
    abort on purpose handle (and clear) the word themselves."""
    yield
    if 'gpu' not in request.keywords or 'src.hipabi' not in sys.modules:
        return
    import torch
    if not torch.cuda.is_available():
        return
    H = sys.modules['src.hipabi']
    culprits = []
    for t, addr, *_rel in H._watch.get(torch.cuda.current_device(), []):
        off = addr - t.data_ptr()
        w = t.view(torch.uint8)[off:off + 4].view(torch.int32)
        if int(w.item()):
            head = t.view(torch.uint8)[off:off + 128].view(torch.int32).tolist() if t.numel() * t.element_size() >= off + 128 else []
            culprits.append('workspace of %d bytes, word at +%d = 0x%x, header %s' % (t.numel() * t.element_size(), off, int(w.item()) & 0xffffffff, head))
    st = H.collect_status()
    v = int(st.item())
    if v:
        st.zero_()
        pytest.fail('a persistent launch of this test raised its abort word (status 0x%x): %s' % (v, '; '.join(culprits)))


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    """On a failing GPU test: the hand-off modes of the decoder's persistent launches (1 = write-through, 2 = XCD-local per
    cluster; status words 26..33 of each work area) go into the report - the first question for any parity flake."""
    outcome = yield
    rep = outcome.get_result()
    if rep.when == 'call' and rep.failed and 'src.functions' in sys.modules:
        try:
            import torch
            F_ = sys.modules['src.functions']
            lines = []
            for (kind, dims, dev), ws in list(F_._DEC_WS.items())[-6:]:
                off = 0
                if kind == 'bwd':
                    continue
                w = ws[off:off + 4096].view(torch.int64).cpu().tolist()
                lines.append('decoder %s work area dims %s: abort %d modes %s consensus %s' % (kind, dims, w[0] & 0xffffffff, w[26:34], [hex(x) for x in w[64:72]]))
            rep.sections.append(('decoder hand-off state', '\n'.join(lines)))
        except Exception as e:       # diagnostics must never mask the failure
            rep.sections.append(('decoder hand-off state', 'unavailable: %r' % (e,)))
