"""The C-ABI library loads and exports every symbol include/legion_amd.h declares (no compute)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "legion_amd.h")
LIB = os.path.join(ROOT, "legion-1_amd", "csrc", "liblegion_amd.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.dirname(LIB), "-j8", "liblegion_amd.so"], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(LIB)


def test_header_declares_the_reference_entry_points():
    names = declared_functions()
    # same names as the reference's extern "C" symbols (src/Kernels.cuh:24-93, *_Storage.cuh, CUDA_IPC_Service.h:35)
    for ref in ("d_alloc_space", "d_alloc_space_managed", "d_copy_2_h", "d_free_space", "SetGPUDevice", "GetGPUDevice",
                "host_alloc_space", "batch_generator_kernel", "GPU_Random_Sampling", "get_feature_kernel",
                "make_update_plan", "update_cache", "NewGPUMemoryGraphStorage", "NewGPUMemoryNodeStorage", "NewIPCEnv",
                "NewBatchGenerator", "NewRandomSampler", "NewFeatureExtractor", "NewCachePlanner", "NewCacheUpdater"):
        assert ref in names
    assert len(names) > 150


def test_library_exports_every_declared_symbol(built_lib):
    missing = [n for n in declared_functions() if not hasattr(built_lib, n)]
    assert not missing, missing


def test_python_binding_table_matches_header(built_lib):
    import legion1_amd.capi as K
    names = set(declared_functions())
    unknown = [n for n in K._SIGS if n not in names]
    assert not unknown, unknown
    assert K.lib().legion_version().startswith(b"legion-amd")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under legion-1_amd/ may reference it."""
    bad = []
    roots = [os.path.join(ROOT, d) for d in ("legion-1_amd", "profiles", "examples", "include")]     # everything that is not tests/, oracle/, bench.py, __graft_entry__.py
    for dirpath, _, files in (t for r in roots for t in os.walk(r)):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"import oracle|from oracle|legion_oracle|liblegion_oracle|lo_run_batch", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import legion1_amd.capi as K
    monkeypatch.setattr(K, "_lib", None)
    monkeypatch.setattr(K, "DEFAULT_LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.delenv("LEGION_LIB", raising=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.lib()


def test_legion_lib_selects_the_library_and_never_falls_back(monkeypatch, tmp_path):
    """$LEGION_LIB: experiments and the sanitizer build point at THEIR library through it; a path that does not
    exist or is relative is an error, never a silent load of the shipped library."""
    import legion1_amd.capi as K
    monkeypatch.setattr(K, "_lib", None)
    monkeypatch.setenv("LEGION_LIB", str(tmp_path / "variant.so"))
    assert K.lib_path() == str(tmp_path / "variant.so")
    with pytest.raises(RuntimeError, match="variant.so is missing"):
        K.lib()
    monkeypatch.setenv("LEGION_LIB", "csrc/liblegion_amd.so")
    with pytest.raises(RuntimeError, match="absolute path"):
        K.lib()
    monkeypatch.setenv("LEGION_LIB", K.DEFAULT_LIB_PATH)          # the shipped library named explicitly: loads, same symbols
    assert K.lib().legion_version().startswith(b"legion-amd")
    monkeypatch.setattr(K, "_lib", None)


def test_c_abi_survives_null_arguments():
    """Every launcher / builder of the boundary called with NULL handles or out-of-range counts (tests/abi_null_args.py, in a child so
    that a crash is a test failure, not the end of the session): each call returns, argument errors are sticky strings, nothing exits."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_null_args.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "survived" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "NewIPCEnv: device_count must be 1..8" in r.stdout and "GPUGraphStorage_Build: null argument" in r.stdout


def test_ipc_service_module_has_the_surface_the_reference_trainers_call():
    """tests/golden/trainer_api.json: every `ipc_service.<name>(...)` call of the reference's three trainer scripts with its argument count and the number of
    values the script unpacks (extracted from their source by the Python parser: `python oracle/make_golden_trainer_api.py`; the scripts need dgl and cannot be
    imported).  The drop-in module must export exactly that surface with those arities; the counts returned at 2 hops (7 tensors, 4 sizes, 3 steps) are checked
    against the served batches in tests/test_gpu_ipc.py."""
    import json
    import sys
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "trainer_api.json")))
    assert g["surface"] == {"finalize": [[0, None]], "get_block_size": [[0, 4]], "get_next": [[1, 7]], "get_steps": [[0, 3]], "initialize": [[0, None]], "synchronize": [[0, None]]}
    assert [len(v) for v in g["scripts"].values()] == [12, 12, 12]      # twelve call sites in each of the three scripts
    sys.path.insert(0, os.path.join(ROOT, "legion-1_amd", "ipc_service"))
    import torch  # noqa: F401  (the extension links libtorch)
    import ipc_service
    for name, sigs in g["surface"].items():
        fn = getattr(ipc_service, name)
        (n_args, _), = sigs
        doc = fn.__doc__.splitlines()[0]                              # pybind11 signature line: name(arg0: int) -> ...
        params = doc[doc.index("(") + 1:doc.index(")")].strip()
        assert (len(params.split(",")) if params else 0) == n_args, (name, doc)
