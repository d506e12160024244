"""Repository rules: the product never imports the oracle, ships no CPU fallback, and nothing
that runs on the GPU box reads /root/reference."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(sub):
    for d, _, files in os.walk(os.path.join(ROOT, sub)):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h', '.cpp')):
                yield os.path.join(d, f)


def test_product_never_touches_the_oracle():
    pat = re.compile(r'^\s*(from|import)\s+(oracle|tests)\b|libgu_oracle|oracle/_build', re.M)
    for path in list(_py_files('griduniverse_amd')) + list(_py_files('include')) + list(_py_files('examples')) + \
            list(_py_files('compat')) + list(_py_files('tools')):
        assert not pat.search(open(path).read()), path + ' references the oracle'


def test_oracle_use_in_entry_points_is_confined():
    """bench.py and its parts (benchlib/) may import oracle only inside cpu_baseline* functions: the cpu_baseline leg
    (benchlib/cpu_leg.py) and the checks of what the GPU produced (benchlib/checks.py) -- never in what is measured."""
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', flags=re.M)
    seen = 0
    for path in [os.path.join(ROOT, 'bench.py')] + sorted(_py_files('benchlib')):
        text = open(path).read()
        allowed = []
        for m in re.finditer(r'^def (cpu_baseline\w*)\(', text, flags=re.M):
            end = text.find('\ndef ', m.start() + 1)
            allowed.append((m.start(), end if end > 0 else len(text)))
        imports = [m.start() for m in pat.finditer(text)]
        seen += len(imports)
        assert all(any(a < i < b for a, b in allowed) for i in imports), path
        if imports:
            assert path.endswith(('cpu_leg.py', 'checks.py')), path
    assert seen


# the fixture generators: run only in the build container, where the reference is mounted
GENERATORS = ('tests/golden/make_golden.py', 'tests/golden/calibrate_cpu.py', 'tests/golden/reference_api_latency.py',
              'tests/test_reference_pin.py')  # (the last one RUNS make_golden.py --check where the reference is, and skips elsewhere)


def test_the_reference_pin_check_is_not_a_gpu_test():
    text = open(os.path.join(ROOT, 'tests', 'test_reference_pin.py')).read()
    assert 'mark.gpu' not in text and 'skipif' in text  # never selected by `-m gpu`, skipped where /root/reference is absent


def test_nothing_on_the_gpu_box_reads_the_reference():
    for path in list(_py_files('griduniverse_amd')) + list(_py_files('tests')) + list(_py_files('oracle')) + \
            list(_py_files('tools')) + [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')]:
        if path.endswith('test_layout.py') or path.replace(os.sep, '/').endswith(GENERATORS):
            continue
        assert '/root/reference' not in open(path).read(), path


def test_only_tests_bench_and_smoke_touch_the_oracle():
    """oracle/ may be imported from tests/ (incl. the fixture generators), from smoke() and from bench.py's
    cpu_baseline leg -- nowhere else (tools/, examples/, the package)."""
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for sub in ('tools', 'examples', 'griduniverse_amd', 'include'):
        for path in _py_files(sub):
            assert not pat.search(open(path).read()), path
    entry = open(os.path.join(ROOT, '__graft_entry__.py')).read()
    assert all(m.start() > entry.index('def smoke') for m in pat.finditer(entry))


def test_oracle_header_declares_test_infrastructure():
    assert 'TEST INFRASTRUCTURE ONLY' in open(os.path.join(ROOT, 'oracle', '__init__.py')).read()
    assert 'TEST INFRASTRUCTURE ONLY' in open(os.path.join(ROOT, 'oracle', 'gu_oracle.c')).read()
