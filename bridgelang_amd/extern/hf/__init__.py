"""bridgelang_amd.extern.hf: part of the MI355X-native OpenVLA path."""
