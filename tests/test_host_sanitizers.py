"""ThreadSanitizer and AddressSanitizer + UBSan runs of the threaded host side of the library (csrc/host_rng.c,
host_graph.c, host_eig.c) under its stress driver tests/native/host_stress.c: the draw thread with a follow-up and a
concurrent draw on the calling thread, the multi-threaded cluster order / hash / copy, the eigen-solver from four threads,
fork after use.  `make -C cna_amd/csrc tsan asan` is the same thing by hand; profiles/r05_{tsan,asan}.txt keep a log.
(What round 5's first runs found: memcpy(_, NULL, 0) in the cluster order's frontier merge, and a per-thread work space
of the eigen-solver that outlived its thread.)"""
import os
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cna_amd', 'csrc')


@pytest.mark.parametrize('target,marks', [('tsan', ('WARNING: ThreadSanitizer', 'ThreadSanitizer: data race')),
                                          ('asan', ('ERROR: AddressSanitizer', 'ERROR: LeakSanitizer', 'runtime error:'))])
def test_host_stress_under_sanitizer(target, marks):
    if shutil.which('gcc') is None and shutil.which('cc') is None:
        pytest.skip('no C compiler')
    p = subprocess.run(['make', '-C', CSRC, target], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode(errors='replace')
    if 'cannot find -ltsan' in out or 'cannot find -lasan' in out or 'unrecognized' in out:
        pytest.skip('sanitizer runtime not installed')
    for m in marks:
        assert m not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]
    assert 'host_stress ok' in out
