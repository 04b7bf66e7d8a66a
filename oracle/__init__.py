"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/xr_oracle.h)."""
