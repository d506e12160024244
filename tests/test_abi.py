"""The C-ABI library loads on a machine without a GPU, exports every symbol include/gu.h
declares, and fails loudly (never falls back) when asked to compute without a device."""
import os
import re

import pytest

from griduniverse_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name=None):
    """Entry points declared in include/gu.h (the surface a reference maintainer binds) and include/gu_diag.h (introspection and
    measurement aids), or in one of the two."""
    out = []
    for header in ([name] if name else ['gu.h', 'gu_diag.h']):
        text = open(os.path.join(ROOT, 'include', header)).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        out += re.findall(r'^\s*int\s+(gu_\w+)\s*\(', text, flags=re.M)
    assert len(out) == len(set(out)), 'an entry point is declared twice'
    return sorted(out)


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_binding_and_library_agree(lib):
    declared = header_symbols()
    assert len(declared) >= 30
    assert declared == sorted(_lib.SIGNATURES), 'python binding table differs from include/gu.h + include/gu_diag.h'
    for name in declared:
        assert hasattr(lib, name), 'libgu.so does not export ' + name
    # the split: what a maintainer of the reference binds, and what only bench.py / tools / tests look at
    surface, diag = header_symbols('gu.h'), header_symbols('gu_diag.h')
    assert len(surface) >= 45 and 10 <= len(diag) <= 25
    for name in ('gu_create', 'gu_set_grid', 'gu_set_grids', 'gu_seed', 'gu_reset', 'gu_step', 'gu_step_device', 'gu_rollout', 'gu_get_state', 'gu_set_state',
                 'gu_done_indices', 'gu_vi_sweep', 'gu_allgather_view', 'gu_last_error', 'gu_version', 'gu_destroy'):
        assert name in surface, name
    for name in diag:
        assert re.match(r'gu_(timer_|rollout_pac|vi_last_|vi_xcd_torn|trajectory_placement|probe_trajectory)', name), name + ' does not look like introspection'


def test_version_and_error_text(lib):
    assert lib.gu_version() == 1
    assert isinstance(_lib.last_error(), str)


def test_gfx950_code_object_is_embedded():
    blob = open(_lib.LIB_PATH, 'rb').read() if os.path.exists(_lib.LIB_PATH) else b''
    assert b'gfx950' in blob, 'libgu.so carries no gfx950 code object'
    assert b'gu_rollout_kernel' in blob and b'gu_step_kernel' in blob and b'gu_vi_sweep_step_kernel' in blob


@pytest.mark.skipif(_lib.os.path.exists('/dev/kfd'), reason='a GPU is present; the no-device path is not reachable')
def test_compute_without_gpu_fails_loudly(lib):
    import griduniverse_amd as gua
    assert _lib.device_count() == 0
    with pytest.raises(gua.GuError):
        gua.GridUniverseEnv().step(1)
    with pytest.raises(gua.GuError):
        gua.VecGridUniverse(8)
    with pytest.raises(gua.GuError):
        gua.GridUniverseEnv().look_step_ahead(0, 1)


def test_library_carries_the_hash_of_the_sources_it_was_built_from(lib, tmp_path, monkeypatch):
    """libgu.so travels to the GPU box next to its sources; the binding refuses one built from other sources."""
    assert _lib.built_hash() == _lib.source_hash() and not _lib.is_stale()
    assert re.fullmatch(r'[0-9a-f]{16}', _lib.built_hash())
    # a source tree that differs (one byte appended to a copy of one file) yields another hash -> load() would refuse
    import shutil
    fake = tmp_path / 'csrc'
    shutil.copytree(_lib.CSRC, fake)
    with open(fake / 'gu_rng.hpp', 'a') as f:
        f.write('\n')
    monkeypatch.setattr(_lib, 'CSRC', str(fake))
    assert _lib.source_hash() != _lib.built_hash() and _lib.is_stale()
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(_lib.GuError) as err:
        _lib.load()
    assert 'built from other sources' in str(err.value)
    monkeypatch.setenv('GU_ALLOW_STALE_LIB', '1')
    assert _lib.load() is not None


def test_stale_library_then_build_then_load_in_one_process(tmp_path, monkeypatch):
    """Looking at the hash of a present-but-stale libgu.so must not map it: otherwise the rebuilt file could never be loaded by
    the process that rebuilt it (glibc answers a second dlopen of the same path with the image it already holds)."""
    import shutil
    good = open(_lib.LIB_PATH, 'rb').read()
    marker = ('GU_SRCHASH=%s;' % _lib.source_hash()).encode()
    assert good.count(marker) == 1
    stale = tmp_path / 'libgu.so'
    stale.write_bytes(good.replace(marker, b'GU_SRCHASH=0123456789abcdef;'))  # "built from other sources"
    monkeypatch.setattr(_lib, 'LIB_PATH', str(stale))
    monkeypatch.setattr(_lib, '_lib', None)
    assert _lib.built_hash() == '0123456789abcdef' and _lib.is_stale()
    assert _lib.build_marker() == ('0123456789abcdef', 'product')
    with pytest.raises(_lib.GuError):
        _lib.load()  # refused: stale (and refusing did not leave the stale image behind as the loaded library)
    assert _lib._lib is None
    shutil.copyfile(os.path.join(ROOT, 'griduniverse_amd', 'lib', 'libgu.so'), str(stale) + '.new')  # "the rebuild"
    os.replace(str(stale) + '.new', str(stale))
    assert not _lib.is_stale()
    lib = _lib.load()
    buf = _lib.ctypes.create_string_buffer(64)
    lib.gu_source_hash(buf, 64)
    assert buf.value.decode() == _lib.source_hash()


def test_options_have_defaults_ranges_and_process_wide_values(lib):
    for name, builtin in (('rollout_block', 256), ('rollout_rows', -1), ('rollout_multi', -1), ('vi_path', 0), ('mc_scratch_mb', 2048),
                          ('traj_candidates', 4), ('traj_far_candidates', 0), ('rollout_pace', -1), ('traj_stride_mib', 3072), ('traj_far_mib', 49152),
                          ('step_sync', 0), ('rollout_xcd', 0), ('traj_probe_all', 0), ('vi_xcd_block', 0), ('pace_target', 7200), ('pace_bar_num', 20),
                          ('pace_gain_q', 128), ('pace_dec_q', 8), ('traj_layout', -1), ('pace_record', 1), ('pace_probe_every', 1024), ('pace_adapt', 1), ('rollout_half_waves', -1), ('rollout_entry', 1), ('sync_spin_us', 5000)):
        assert _lib.get_default_option(name) == builtin, name
    _lib.set_default_option('rollout_block', 512)
    try:
        assert _lib.get_default_option('rollout_block') == 512
    finally:
        _lib.set_default_option('rollout_block', None)
    assert _lib.get_default_option('rollout_block') == 256
    for name, bad in (('rollout_block', 100), ('rows_copies', 3), ('rollout_multi_k', 3), ('vi_path', 9), ('traj_candidates', 0), ('vi_xcd_block', 128)):
        with pytest.raises(_lib.GuError):
            _lib.set_default_option(name, bad)
    assert sorted(v for v in _lib.OPTIONS.values() if v < 100) == list(range(1, 31))


def test_the_product_library_has_no_code_for_the_unsafe_experiments(lib, monkeypatch):
    """Round 2 shipped GU_TRAJ_UNCACHED=1 (a mode documented to make readers of the trajectory see stale bytes) one environment
    variable away.  The product library now ignores the variable, refuses the option, and does not even import the allocator the
    mode needs; only `make exp` (libgu_exp.so, tools/) carries it."""
    import subprocess
    monkeypatch.setenv('GU_TRAJ_UNCACHED', '1')
    monkeypatch.setenv('GU_TRAJ_POISON', '1')
    monkeypatch.setenv('GU_MC_POISON', '1')
    for name in ('x_traj_uncached', 'x_traj_poison', 'x_mc_poison'):
        assert _lib.get_default_option(name) == 0
        with pytest.raises(_lib.GuError) as err:
            _lib.set_default_option(name, 1)
        assert err.value.code == -6
    assert _lib.build_marker()[1] == 'product'
    undefined = subprocess.run(['nm', '-D', '--undefined-only', _lib.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert 'hipMalloc' in undefined and 'hipExtMallocWithFlags' not in undefined
    # the environment-only switches of rounds 1 and 2 are gone from the product library altogether
    blob = open(_lib.LIB_PATH, 'rb').read()
    for gone in (b'GU_VI_MULTI_LAUNCH', b'GU_VI_CLUSTER', b'GU_TRAJ_DEBUG', b'GU_TRAJ_STRIDE_GIB', b'GU_TRAJ_FAR_GIB'):
        assert gone not in blob, gone
    assert b'GU_RCCL_LIB' in blob and b'GU_DEBUG' in blob  # the two process-level variables it does read
