from .metrics_protocols import (  # noqa: F401
    AccuracyScore, AucScore, F1Score, LogLossScore, MetricEvaluator, MrrScore, NdcgScore, RootMeanSquaredError,
)
