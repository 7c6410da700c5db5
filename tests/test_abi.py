"""CPU: the C ABI library loads and exports every symbol include/nova_hip.h declares; the ctypes
table in nova_pointcloud_amd/hip.py covers the same set; the loader fails loudly when the library is absent."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nova_hip.h")


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return set(re.findall(r"\b(nova_[a-z0-9_]+)\s*\(", text))


@pytest.fixture(scope="module")
def built_lib():
    from nova_pointcloud_amd import hip

    if not os.path.exists(hip.lib_path()):
        subprocess.run(["make", "-C", os.path.join(ROOT, "nova_pointcloud_amd", "csrc"), "-j4"], check=True)
    return hip


def test_header_symbols_are_exported(built_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib.lib_path()], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (nova_[a-z0-9_]+)", out))
    want = declared_symbols()
    assert len(want) >= 18
    assert want <= exported, sorted(want - exported)


def test_ctypes_table_matches_header(built_lib):
    table = set(built_lib.SIGNATURES) | set(built_lib.PLAIN)
    assert table == declared_symbols(), sorted(table ^ declared_symbols())
    lib = built_lib.load(check_device=False)   # binds every symbol; no compute without a GPU
    assert lib.nova_version() == int(re.search(r"#define NOVA_HIP_VERSION (\d+)", open(HEADER).read()).group(1))
    assert lib.nova_version() == built_lib.ABI_VERSION  # what hip.load() insists on: a stale library fails loudly instead of mis-binding


def test_every_entry_point_cites_the_reference():
    text = open(HEADER).read()
    for needle in ("vision_transformer.py", "diffusion_mlp.py", "normalization.py", "embeddings.py", "guidance_scaler.py",
                   "scheduling_cfm.py", "transformer_3d.py"):
        assert needle in text


def test_missing_library_raises(monkeypatch):
    import nova_pointcloud_amd.hip as H

    monkeypatch.setattr(H, "_lib", None)
    monkeypatch.setattr(H, "_LIB_PATH", "/nonexistent/libnova_hip.so")
    with pytest.raises(H.NovaHipError, match="no CPU fallback"):
        H.load(check_device=False)


def test_dtype_codes_follow_the_header():
    """NOVA_F32 / NOVA_BF16 / NOVA_F16 of include/nova_hip.h and the binding's torch dtype map (float16 is the default
    precision of every caller of the reference: scripts/app_nova_t2i.py:36)."""
    import re

    import torch

    from nova_pointcloud_amd import hip

    enum = re.search(r"typedef enum \{([^}]*)\} nova_dtype;", open(HEADER).read()).group(1)
    codes = {k.strip(): int(v) for k, v in (item.split("=") for item in enum.split(","))}
    assert codes == {"NOVA_F32": 0, "NOVA_BF16": 1, "NOVA_F16": 2}
    assert [hip.dtype_code(t) for t in (torch.float32, torch.bfloat16, torch.float16)] == [0, 1, 2]
    with pytest.raises(hip.NovaHipError):
        hip.dtype_code(torch.float64)
