"""sage_amd: MI355X-native ablation hot path for SaGe 2.0."""
