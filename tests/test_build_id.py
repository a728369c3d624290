"""The product's kernel header carries no experiment scaffolding, and a build's id covers its flags.

Round 5's `csrc/kernels_fft.hpp` had 40 preprocessor conditionals over 35 `IYK_FFT_*` knobs, some of which compute wrong results
on purpose (timing-only builds), and the build id hashed sources only — a timing-only library carried the product's id.  Now the
knobs live in `tools/experiments/kernels_fft_r05_knobs.hpp`, reachable only through `-DIYK_EXPERIMENT_KERNELS_FFT=...`, and
`tools/src_hash.py` hashes the compiler flags as well."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_product_fft_header_has_no_knobs():
    text = open(os.path.join(ROOT, "iyokan_amd", "csrc", "kernels_fft.hpp")).read()
    conditionals = re.findall(r"^\s*#\s*(?:if|ifdef|ifndef|elif)\b", text, flags=re.M)
    assert len(conditionals) <= 8, conditionals
    assert "IYK_FFT_TIMING" not in text and "IYK_FFT_TRACE" not in text and "IYK_LATFFT_TRACE" not in text
    for name in ("blind_rotate_fft.hpp", "fft512.hpp"):
        other = open(os.path.join(ROOT, "iyokan_amd", "csrc", name)).read()
        assert not re.search(r"IYK_FFT_(DIFF_PLAIN|DIFF_R04|LF_FENCE)", other), name
    # the A/B version still exists, outside the product and outside its hash
    assert os.path.exists(os.path.join(ROOT, "tools", "experiments", "kernels_fft_r05_knobs.hpp"))


def test_build_id_covers_flags():
    import src_hash

    product = src_hash.build_id()
    assert re.fullmatch(r"[0-9a-f]{16}", product)
    assert src_hash.build_id([]) == product
    assert src_hash.build_id(["-O3", "-o", "x.so"]) == product                 # flags that do not change the code
    timing = src_hash.build_id(["-DIYK_FFT_TIMING_NOKEYS"])
    assert timing != product
    assert src_hash.build_id(["-DIYK_FFT_TIMING_NOKEYS", "-DIYK_FFT_KH_DEPTH=6"]) not in (product, timing)
    # order of the flags is irrelevant
    assert src_hash.build_id(["-DB=1", "-DA=2"]) == src_hash.build_id(["-DA=2", "-DB=1"])
    # the command-line form the scripts use
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "src_hash.py"), "-DIYK_FFT_TIMING_NOKEYS"],
                         capture_output=True, text=True, check=True).stdout
    assert out == timing


def test_loaded_library_reports_the_product_id():
    lib = os.path.join(ROOT, "iyokan_amd", "lib", "libiyokan_hip.so")
    if not os.path.exists(lib):
        import pytest

        pytest.skip("library not built")
    import src_hash

    blob = open(lib, "rb").read()
    assert src_hash.build_id().encode() + b"\0" in blob        # exactly the id, not id + "+x"
    assert src_hash.build_id().encode() + b"+x" not in blob
