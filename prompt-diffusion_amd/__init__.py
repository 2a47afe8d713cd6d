"""pdengine: MI355X-native Prompt-Diffusion DDIM sampling engine (host side).

The compute path is ``csrc/libpdengine.so`` (hand-written gfx950 HIP kernels behind
the C ABI of ``include/pdengine.h``); this package is the Python host that mirrors
the reference's ``PromptDiffusionPipeline.__call__`` (pipeline_prompt_diffusion.py:890)
and ``DDIMSampler.sample`` (cldm/ddim_hacked.py:55) surfaces over ctypes.
"""
__version__ = "0.1.0"
