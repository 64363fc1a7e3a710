"""pip install -e .   (from the repository root, ROCm 7.x image with hipcc and PyTorch-ROCm)

Replaces the two install steps of zzz5y/street_crafter that bring in the CUDA operators
(README.md:35 `pip install "git+https://github.com/dendenxu/gsplat.git"` and `pip install ./submodules/simple-knn`):
the packages `gsplat`, `simple_knn` and `street_crafter_amd` of this repository are installed under those names and
the HIP library + the compiled binding layer are built IN-TREE (street_crafter_amd/lib/), which is where the
packages load them from -- an editable install is the intended form.  `python -m street_crafter_amd.build` does the
same build without installing anything; putting the repository root on PYTHONPATH is all the packages need.
"""
import os
import sys

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

HERE = os.path.dirname(os.path.abspath(__file__))


def _build_native():
    sys.path.insert(0, HERE)
    from street_crafter_amd import build as b
    print("built", b.build(verbose=False))
    print("built", b.build_binding(verbose=False))


class BuildPy(build_py):
    def run(self):
        _build_native()
        super().run()


class Develop(develop):
    def run(self):
        _build_native()
        super().run()


setup(
    name="street_crafter_amd",
    version="0.4.0",
    description="MI355X (gfx950) HIP implementation of the gsplat / simple_knn operators StreetCrafter calls",
    packages=find_packages(include=["street_crafter_amd", "street_crafter_amd.*", "gsplat", "gsplat.*", "simple_knn",
                                    "simple_knn.*"]),
    package_data={"street_crafter_amd": ["lib/*.so", "csrc/*", "../include/*.h"]},
    python_requires=">=3.9",
    install_requires=[],          # torch (PyTorch-ROCm) and numpy come with the image; nothing is fetched
    cmdclass={"build_py": BuildPy, "develop": Develop},
    zip_safe=False,
)
