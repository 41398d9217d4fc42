"""Import-path compatibility: `cvap.module`, `cvap.model`, `cvap.monitor`, `cvap.util` resolve to the MI355X-native
implementations in `vipant_amd`, so code written against the reference's operator API runs unchanged."""
