"""pip install . -- builds the gfx950 library and the torch adapter IN-TREE (make -C gaustudio_amd/csrc) and installs
`gaustudio_amd` plus the drop-in import name `gaustudio_diff_gaussian_rasterization`, i.e. what
`pip install submodules/gaustudio-diff-gaussian-rasterization` gives the reference (its setup.py:13-33, CUDAExtension).
Needs hipcc (ROCm 7.x) and PyTorch-ROCm; no other dependency."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithNative(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gaustudio_amd", "csrc"), "-j4"])
        super().run()


setup(
    name="gaustudio-amd",
    version="0.1.0",
    description="MI355X-native (gfx950) differentiable 3D-Gaussian rasterizer, drop-in for gaustudio_diff_gaussian_rasterization",
    packages=["gaustudio_amd", "gaustudio_diff_gaussian_rasterization"],
    package_data={"gaustudio_amd": ["*.so"]},
    include_package_data=True,
    cmdclass={"build_py": BuildWithNative},
    python_requires=">=3.10",
)
