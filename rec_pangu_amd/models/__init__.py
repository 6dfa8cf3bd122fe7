from . import layers, ranking, multi_task  # noqa: F401
