"""CPU oracle for the BCn transform hot path -- TEST INFRASTRUCTURE ONLY (see dxtlt_oracle.h)."""
