"""CPU oracle (test infrastructure only): see ilcc_oracle.h. Import ``oracle.binding``."""
