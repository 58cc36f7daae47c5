"""bridgelang_amd.util: part of the MI355X-native OpenVLA path."""
