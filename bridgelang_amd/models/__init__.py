"""bridgelang_amd.models: part of the MI355X-native OpenVLA path."""
