# The invariants the reference's vignette asserts with stopifnot() (vignettes/bamsignals.Rmd:134,
# 148, 150, 152, 224, 235) as testthat expectations, on the package's own example data.
context("vignette invariants")
library(GenomicRanges)

test_that("the relations between bamCount, bamProfile, ss and binning hold", {
    bampath <- system.file("extdata", "randomBam.bam", package="bamsignals")
    genes <- get(load(system.file("extdata", "randomAnnot.Rdata", package="bamsignals")))
    sigs <- bamProfile(bampath, genes, verbose=FALSE)
    counts <- bamCount(bampath, genes, verbose=FALSE)
    expect_equal(length(sigs), length(genes))
    expect_true(all(width(sigs) == width(genes)))                        # Rmd:134
    expect_equal(sapply(as.list(sigs), sum), counts)                     # a profile sums to the count
    sssigs <- bamProfile(bampath, genes, verbose=FALSE, ss=TRUE)
    expect_true(all(colSums(sssigs[1]) == sigs[1]))                      # Rmd:148
    expect_true(all(width(sssigs) == width(sigs)))                       # Rmd:150
    expect_equal(length(sssigs[1]), 2 * length(sigs[1]))                 # Rmd:152
    expect_equal(rownames(sssigs[1]), c("sense", "antisense"))
    proms <- GenomicRanges::promoters(genes, upstream=100, downstream=100)
    psigs <- bamProfile(bampath, proms, ss=FALSE, verbose=FALSE)
    pss <- bamProfile(bampath, proms, ss=TRUE, verbose=FALSE)
    m <- alignSignals(psigs); a <- alignSignals(pss)
    expect_equal(dim(m), c(200L, length(proms)))
    expect_equal(dim(a), c(2L, 200L, length(proms)))
    expect_true(all(m == a["sense",,] + a["antisense",,]))               # Rmd:199
    binsize <- 20
    b <- bamProfile(bampath, proms, binsize=binsize, verbose=FALSE)
    expect_true(all(width(b) == ceiling(width(psigs) / binsize)))        # Rmd:224
    expect_equal(colSums(matrix(rowMeans(m), nrow=binsize)), rowMeans(alignSignals(b)))   # Rmd:235
    cov <- bamCoverage(bampath, genes, verbose=FALSE)
    expect_true(all(width(cov) == width(genes)))
})
