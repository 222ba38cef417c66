"""CPU: the product never reaches the oracle and has no CPU fallback.

* nothing under pushworld_amd/ imports (or even names) the oracle package: only tests/, __graft_entry__.smoke() and
  bench.py's cpu_baseline legs may;
* without the HIP library the package fails at import with an ImportError that says so."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_sources_do_not_reference_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|\boracle\.(pw_oracle|c_oracle|py_bench)\b|libpw_oracle", re.M)
    offenders = []
    for base, _, files in os.walk(os.path.join(ROOT, "pushworld_amd")):
        for name in files:
            if name.endswith((".py", ".cpp", ".hip", ".inc", ".h")):
                with open(os.path.join(base, name), errors="replace") as f:
                    if pat.search(f.read()):
                        offenders.append(os.path.join(base, name))
    assert not offenders, offenders
    # bench.py and the configuration suite reach the oracle only through tools/cpu_baselines.py (the cpu_baseline leg),
    # imported inside the CPU-baseline code paths -- never at module level, never from the timed GPU loops
    for rel in ("bench.py", os.path.join("tools", "config_suite.py")):
        with open(os.path.join(ROOT, rel)) as f:
            src = f.read()
        assert not pat.search(src), rel
        for m in re.finditer(r"^(\s*)from tools(\.cpu_baselines| import cpu_baselines)", src, re.M):
            assert len(m.group(1)) >= 4, (rel, "cpu_baselines must be imported inside a function")
            head = src[:m.start()]
            fn = re.findall(r"^def (\w+)\(", head, re.M)[-1]
            assert fn in ("cpu_baseline", "python_env_baseline", "run_c1", "run_c2", "run_c3", "run_c4", "run_c5"), (rel, fn)
            if rel != "bench.py":  # ... and there behind the `if cpu:` switch
                assert "if cpu:" in head[head.rindex("def " + fn):], (rel, fn)


def test_missing_library_is_an_import_error_not_a_fallback(tmp_path):
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import os\n"
        "os.environ['PUSHWORLD_AMD_LIB'] = %r\n"
        "try:\n"
        "    import pushworld_amd._capi\n"
        "except ImportError as e:\n"
        "    assert 'no CPU fallback' in str(e), e\n"
        "    print('IMPORT_ERROR_OK')\n"
    ) % (ROOT, str(tmp_path / "nowhere" / "libpushworld_amd.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "IMPORT_ERROR_OK" in out.stdout, out.stdout + out.stderr
