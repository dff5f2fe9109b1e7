from .metrics import clustering_coef, degree_histogram  # noqa: F401
