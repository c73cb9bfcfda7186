"""7B ("gla_swa") model family: registration hook
(reference: ``vllm_hymeta/model_for_7B/__init__.py:8-13``)."""


def register_7B_model() -> None:
    from transformers import AutoConfig
    from vllm import ModelRegistry

    from .configuration_gla_swa import GLAswaConfig

    try:
        AutoConfig.register("gla_swa", GLAswaConfig)
    except ValueError:
        pass
    ModelRegistry.register_model("GLAswaForCausalLM", "vllm_hymeta.model_for_7B.modeling_gla_swa:GLAswaForCausalLM")
