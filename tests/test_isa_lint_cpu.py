"""ISA lint on the CPU box (hipcc cross-compiles gfx950 without a GPU): the one source that stores through buffer
descriptors must not contain a `buffer_store_dwordx3/4` with a REGISTER scalar offset whose data registers the next
vector instruction overwrites -- hipcc (ROCm 7.2) inserts no wait state for that pair and gfx950 needs one
(tools/vmem_store_war_lint.py; DESIGN.md 7.0a, "a second hazard")."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qpwcnet_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _sources_with_buffer_stores():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") and "raw_buffer_store" in open(os.path.join(CSRC, f)).read():
            out.append(f)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_buffer_store_with_register_offset_is_followed_by_a_write_of_its_data(tmp_path):
    srcs = _sources_with_buffer_stores()
    assert srcs, "the fp32 SeparableConv2D epilogue stores through a buffer descriptor"
    procs = []
    for f in srcs:
        s = str(tmp_path / (f + ".s"))
        procs.append((f, s, subprocess.Popen(
            [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
             "-S", "--cuda-device-only", os.path.join(CSRC, f), "-o", s],
            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    for f, s, p in procs:
        assert p.wait(timeout=900) == 0, "hipcc -S failed for " + f
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "vmem_store_war_lint.py"), s],
                           capture_output=True, text=True)
        assert r.returncode == 0, f + ":\n" + r.stdout[-2000:]


def test_the_lint_sees_the_pair(tmp_path):
    s = tmp_path / "bad.s"
    s.write_text("_Z3badv:\n\tbuffer_store_dwordx4 v[4:7], v29, s[0:3], s4 offen\n\tv_pk_mul_f32 v[4:5], v[10:11], v[12:13]\n"
                 "\tbuffer_store_dwordx4 v[4:7], v29, s[0:3], 0 offen offset:192\n\tv_mov_b32_e32 v4, 0\n\ts_endpgm\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "vmem_store_war_lint.py"), str(s)],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "suspects: 1" in r.stdout, r.stdout
