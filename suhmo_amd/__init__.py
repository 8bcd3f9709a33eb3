"""suhmo_amd -- MI355X-native hydraulic-head solve (hot path of EnnaDelfen/SUHMO)."""
