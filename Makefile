# Top-level Makefile: the reference's targets (Makefile:60-67) `spmv` and `spgemm` produce bmsparse_spmv_float and
# bmsparse_spgemm_float here; `lib` builds the C-ABI library they link; `oracle` builds the CPU parity checker.
PKG = bmsparse-spgemm-spmv_amd
all: lib spmv spgemm
lib spmv spgemm:
	$(MAKE) -C $(PKG) $@
oracle:
	$(MAKE) -C oracle all
clean:
	$(MAKE) -C $(PKG) clean
	$(MAKE) -C oracle clean
.PHONY: all lib spmv spgemm oracle clean
