from .model import GraphVAE, deg_sim, vec_to_adj  # noqa: F401
