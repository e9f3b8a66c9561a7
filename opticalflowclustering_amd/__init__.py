"""MI355X-native dense-Farneback-flow -> k-means hot path behind the reference's own entry points
(menmitsu/opticalFlowClustering, k-means-color-clustering/).  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"
