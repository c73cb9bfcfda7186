"""vllm_hymeta: MI355X-native drop-in for the FAndromedA/vllm-hymeta plugin surface.

Entry points (same names as the reference's ``setup.py:71-78``):

* ``vllm.platform_plugins``: ``hymeta = vllm_hymeta:register``
* ``vllm.general_plugins``:  ``hymeta_model = vllm_hymeta:register_model``

Both run in every vLLM process before the device is initialised, so importing
this package touches neither torch.cuda nor the HIP library; the kernels are
bound lazily on first use (``vllm_hymeta._C.load``).
"""


def register():
    """Platform plugin hook (reference: ``vllm_hymeta/__init__.py:2-4``).

    Returns the dotted path of the platform class vLLM should instantiate.
    """
    return "vllm_hymeta.platform.HymetaRocmPlatform"


def register_model():
    """General plugin hook (reference: ``vllm_hymeta/__init__.py:6-11``)."""
    from .model_for_7B import register_7B_model
    register_7B_model()

    from .models import register_model as _register_70b
    _register_70b()
