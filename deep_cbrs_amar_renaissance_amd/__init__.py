"""MI355X-native GNN-propagation + hybrid-scoring hot path behind the Deep_CBRS_Amar_Renaissance interface."""
__version__ = "0.1.0"
