"""bridgelang_amd.extern: part of the MI355X-native OpenVLA path."""
