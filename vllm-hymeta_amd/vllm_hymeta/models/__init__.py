"""70B ("hybrid") model family: registration hook (reference: ``vllm_hymeta/models/__init__.py:8-13``)."""


def register_model() -> None:
    from transformers import AutoConfig
    from vllm import ModelRegistry

    from .configuration_hymeta import HymetaConfig

    try:
        AutoConfig.register("hybrid", HymetaConfig)
    except ValueError:
        pass  # already registered: the hook runs in every vLLM process and may run twice
    ModelRegistry.register_model("HymetaForCausalLM", "vllm_hymeta.models.modeling_hymeta:HymetaForCausalLM")
