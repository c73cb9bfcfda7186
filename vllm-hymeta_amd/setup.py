"""Packaging of the MI355X-native vllm_hymeta plugin.

`pip install -e vllm-hymeta_amd` after `make -C vllm-hymeta_amd` (the shared library is built by
the Makefile, in-tree, and shipped as package data).  Entry-point group names and targets are the
ones vLLM's plugin loader looks up (reference: setup.py:71-78).
"""
from setuptools import find_packages, setup

setup(
    name="vllm_hymeta",
    version="0.1.0",
    description="Hymeta hybrid-attention plugin for vLLM on AMD MI355X (gfx950): hand-written HIP kernels",
    packages=find_packages(include=["vllm_hymeta", "vllm_hymeta.*"]),
    package_data={"vllm_hymeta": ["libhymeta_hip.so"]},
    python_requires=">=3.10",
    install_requires=["torch", "transformers"],
    entry_points={
        "vllm.platform_plugins": ["hymeta = vllm_hymeta:register"],
        "vllm.general_plugins": ["hymeta_model = vllm_hymeta:register_model"],
    },
)
