"""CPU oracle for the PLM hot path -- TEST INFRASTRUCTURE ONLY (see plm_oracle.c)."""
