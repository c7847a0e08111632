from .trainer import BaselineTrainer, inference_on_dataset  # noqa: F401
