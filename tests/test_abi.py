"""The C-ABI library loads on a machine without a GPU, exports every symbol include/gu.h
declares, and fails loudly (never falls back) when asked to compute without a device."""
import os
import re

import pytest

from griduniverse_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'gu.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'^\s*int\s+(gu_\w+)\s*\(', text, flags=re.M)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_binding_and_library_agree(lib):
    declared = header_symbols()
    assert len(declared) >= 30
    assert declared == sorted(_lib.SIGNATURES), 'python binding table differs from include/gu.h'
    for name in declared:
        assert hasattr(lib, name), 'libgu.so does not export ' + name


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
