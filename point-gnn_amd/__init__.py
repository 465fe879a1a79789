"""pointgnn_amd -- MI355X-native Point-GNN hot path."""
