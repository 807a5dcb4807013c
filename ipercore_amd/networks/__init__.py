from .params import generator_param_shapes  # noqa: F401
