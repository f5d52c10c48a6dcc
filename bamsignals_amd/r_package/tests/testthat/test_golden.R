# Acceptance test for a box that has R + GenomicRanges + an MI355X: the user API against the
# committed golden vectors (the outputs of the reference's own R test oracle on its fixture BAM,
# over the parameter grid of the reference's tests/testthat/test_methods.R:33-104).
# Prepare the text copies with:  python scripts/export_goldens_for_r.py <dir>
# and point BAMSIGNALS_GOLDEN_DIR at <dir>.
context("golden vectors")
library(GenomicRanges)

gold <- Sys.getenv("BAMSIGNALS_GOLDEN_DIR")

load_expected <- function(path) {
    lines <- readLines(path)
    parts <- strsplit(lines, "\t", fixed=TRUE)
    vals <- lapply(parts, function(p) as.integer(strsplit(p[2], ",", fixed=TRUE)[[1]]))
    names(vals) <- vapply(parts, function(p) p[1], "")
    vals
}

parse_key <- function(key) {
    kv <- strsplit(strsplit(key, "|", fixed=TRUE)[[1]], ",", fixed=TRUE)
    kind <- kv[[1]][1]
    fields <- strsplit(kv[[2]], "=", fixed=TRUE)
    p <- setNames(lapply(fields, function(f) f[2]), vapply(fields, function(f) f[1], ""))
    tf <- if (p$tf == "NULL") NULL else c(50, 200)
    list(kind=kind, shift=as.numeric(p$shift), mapq=as.numeric(p$mapq), ss=identical(p$ss, "1"),
         pe=p$pe, tf=tf)
}

test_that("bamCount / bamProfile / bamCoverage reproduce the golden grid", {
    skip_if(gold == "", "BAMSIGNALS_GOLDEN_DIR not set")
    reg <- read.delim(file.path(gold, "regions.tsv"), stringsAsFactors=FALSE)
    regions <- GRanges(reg$chrom, IRanges(reg$start, width=reg$width), strand=reg$strand)
    plus <- regions; strand(plus) <- "+"
    bampath <- file.path(gold, "randomBam.bam")
    expected <- load_expected(file.path(gold, "expected_grid.txt"))
    for (key in names(expected)) {
        k <- parse_key(key)
        got <- switch(k$kind,
            count = {
                x <- bamCount(bampath, regions, ss=k$ss, shift=k$shift, paired.end=k$pe,
                              mapqual=k$mapq, tlenFilter=k$tf, verbose=FALSE)
                as.integer(x)                      # 2 x n matrix is column-major: sense, antisense, ...
            },
            profile = {
                x <- bamProfile(bampath, regions, ss=k$ss, shift=k$shift, paired.end=k$pe,
                                mapqual=k$mapq, tlenFilter=k$tf, verbose=FALSE)
                expect_equal(width(x), width(regions))
                unlist(lapply(as.list(x), as.integer))
            },
            coverage = {
                x <- bamCoverage(bampath, regions, paired.end=k$pe, mapqual=k$mapq,
                                 tlenFilter=k$tf, verbose=FALSE)
                unlist(as.list(x))
            },
            ff16 = as.integer(bamCount(bampath, plus, ss=FALSE, shift=k$shift, paired.end=k$pe,
                                       mapqual=k$mapq, tlenFilter=k$tf, filteredFlag=16,
                                       verbose=FALSE)))
        expect_equal(got, expected[[key]], label=key)
    }
})

test_that("return shapes match the reference", {
    skip_if(gold == "", "BAMSIGNALS_GOLDEN_DIR not set")
    reg <- read.delim(file.path(gold, "regions.tsv"), stringsAsFactors=FALSE)
    regions <- GRanges(reg$chrom, IRanges(reg$start, width=reg$width), strand=reg$strand)
    bampath <- file.path(gold, "randomBam.bam")
    cnt <- bamCount(bampath, regions, ss=TRUE, verbose=FALSE)
    expect_equal(dim(cnt), c(2L, length(regions)))
    expect_equal(rownames(cnt), c("sense", "antisense"))
    prof <- bamProfile(bampath, regions, ss=TRUE, verbose=FALSE)
    expect_is(prof, "CountSignals")
    expect_equal(dim(prof[1]), c(2L, width(regions)[1]))
    expect_equal(rownames(prof[1]), c("sense", "antisense"))
    expect_error(bamProfile(bampath, regions, binsize=0, verbose=FALSE), "binsize greater or equal to 1")
    expect_warning(bamProfile(bampath, regions, binsize=7, verbose=FALSE), "not a multiple")
    expect_error(bamCount(bampath, GRanges("chrZ", IRanges(1, 10)), verbose=FALSE),
                 "chromosome chrZ not present in the bam file")
})
