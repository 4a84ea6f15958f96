"""CPU oracle (test infrastructure only). See mmlrec_oracle.py."""
