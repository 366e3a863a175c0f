"""Test-suite plumbing.

GPU runs (`-m gpu`) must be able to NAME the test a native crash happened in: round 3's driver run died with SIGABRT and a
faulthandler trailer that pushed everything useful out of the record.  Three things take care of that:

  * before every test its node id goes (flushed) to ``gpurun_out/pytest_gpu_trace.txt``;
  * a tiny monitor child (plain Python, never touches the GPU, started before this process initialises HIP) reads the same ids
    from a pipe; if the pipe closes without the end-of-session mark it prints the last id to the REAL stderr -- after whatever the
    dying process printed, i.e. as the last line of the log;
  * after every GPU test the device is synchronised and HIP's sticky error is read, so an asynchronous fault is raised in the test
    that launched it, not in a later one.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_MONITOR_SRC = r'''
import sys, time
last, done = None, False
for line in sys.stdin:
    line = line.rstrip("\n")
    if line == "DONE":
        done = True
    elif line:
        last = line
if not done:
    time.sleep(0.2)
    sys.stderr.write("\n[rvip-test-monitor] the pytest process ended without finishing its session; last test started: %s\n" % last)
    sys.stderr.flush()
'''

_state = {'trace': None, 'monitor': None}


def _gpu_session(config):
    expr = (config.getoption('markexpr', '') or '').strip()
    return os.environ.get('RVIP_TEST_MONITOR') == '1' or (expr.startswith('gpu') and 'not gpu' not in expr)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    if not _gpu_session(config) or os.environ.get('RVIP_TEST_MONITOR') == '0':
        return
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        _state['trace'] = open(os.path.join(ROOT, 'gpurun_out', 'pytest_gpu_trace.txt'), 'w')
    except OSError:
        _state['trace'] = None
    try:                       # global capture is suspended during pytest_configure: fd 2 is the real stderr here
        _state['monitor'] = subprocess.Popen([sys.executable, '-S', '-E', '-c', _MONITOR_SRC], stdin=subprocess.PIPE,
                                             stdout=subprocess.DEVNULL, stderr=os.dup(2), close_fds=True, text=True)
    except OSError:
        _state['monitor'] = None


def _note(line):
    f = _state['trace']
    if f is not None:
        f.write(line + '\n')
        f.flush()
    m = _state['monitor']
    if m is not None and m.stdin is not None:
        try:
            m.stdin.write(line + '\n')
            m.stdin.flush()
        except (BrokenPipeError, ValueError):
            _state['monitor'] = None


def pytest_runtest_logstart(nodeid, location):
    _note(nodeid)


def pytest_unconfigure(config):
    _note('DONE')
    m, _state['monitor'] = _state['monitor'], None
    if m is not None:
        try:
            m.stdin.close()
            m.wait(timeout=5)
        except Exception:
            pass
    if _state['trace'] is not None:
        _state['trace'].close()
        _state['trace'] = None


@pytest.fixture(autouse=True)
def _device_clean_after_gpu_test(request):
    """A GPU test ends with an idle, error-free device: a kernel fault or a sticky HIP error is raised HERE, in the test that
    caused it."""
    yield
    if request.node.get_closest_marker('gpu') is None:
        return
    import torch
    if not torch.cuda.is_available():
        return
    torch.cuda.synchronize()
    import importlib
    N = importlib.import_module('cmr-landmark-detection_amd._native')
    rc = N.lib().rvip_device_check()       # (our library binds the same HIP runtime as torch: see _native.lib)
    assert rc == 0, 'hipError %d left behind by %s' % (rc, request.node.nodeid)
