from .metrics import aspl, batch_stat, clustering_coef, degree_histogram, degree_l1  # noqa: F401
