from .build import load_inference_source  # noqa: F401
