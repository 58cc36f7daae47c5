"""bridgelang_amd.vla: part of the MI355X-native OpenVLA path."""
