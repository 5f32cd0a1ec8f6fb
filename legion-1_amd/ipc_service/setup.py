"""In-tree build of the trainer-side `ipc_service` module (reference: pytorch_extension/setup.py).
    python setup.py build_ext --inplace
A plain C++ extension: all HIP work happens inside liblegion_amd.so."""
import os

from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CppExtension

here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.normpath(os.path.join(here, "..", "csrc"))
setup(
    name="ipcservice",
    ext_modules=[CppExtension(
        "ipc_service", ["ipc_service.cpp"],
        library_dirs=[csrc], libraries=["legion_amd"],
        extra_link_args=["-Wl,-rpath,$ORIGIN/../csrc"],
        extra_compile_args=["-O2", "-std=c++17"])],
    cmdclass={"build_ext": BuildExtension})
