package io.trino.gpu;

import io.trino.spi.Page;
import io.trino.spi.block.Block;
import io.trino.spi.block.LazyBlock;
import io.trino.spi.connector.ConnectorPageSource;
import io.trino.spi.type.Type;

import java.util.List;

/**
 * The ConnectorPageSource of a ScanFilterAndProjectOperator as the native operator sees it (GpuNative.createScanFilterProject,
 * include/presto_amd.h pa_page_source): the native side pulls pages through nextPage and asks for the LazyBlocks it needs through
 * loadBlock -- the filter's channels first, the projections' other channels only for a page in which the filter selected a position,
 * channels no expression reads never (PageProcessor.java:307-347; ScanFilterAndProjectOperator.java:185-292).  All three methods are
 * called from inside GpuNative.getOutput, on the Driver's thread.
 */
final class GpuPageSource
{
    private final ConnectorPageSource source;
    private final PinnedPagePool staging;
    private Page current;
    private PinnedPagePool.StagedPage staged;

    GpuPageSource(ConnectorPageSource source, List<Type> types)
    {
        this.source = source;
        this.staging = new PinnedPagePool(types);
    }

    /** null = finished; else [positions, channels, slab address, then per channel: type, encoding, values / offsets / nulls offset (-1 = absent), loaded]. */
    long[] nextPage()
    {
        Page page = null;
        while (page == null || page.getPositionCount() == 0) {
            if (source.isFinished()) {
                return null;
            }
            page = source.getNextPage();
        }
        current = page;
        staged = staging.stageLoadedBlocks(page);      // blocks that are not LazyBlocks (or are loaded already); the rest on demand
        long[] out = new long[3 + 6 * staged.channels];
        out[0] = staged.positions;
        out[1] = staged.channels;
        out[2] = GpuNative.bufferAddress(staged.buffer);
        for (int c = 0; c < staged.channels; c++) {
            Block block = page.getBlock(c);
            boolean loaded = !(block instanceof LazyBlock) || block.isLoaded();
            int at = 3 + 6 * c;
            out[at] = staged.types[c];
            out[at + 1] = staged.encodings[c];
            out[at + 2] = loaded ? staged.valueOffsets[c] : -1;
            out[at + 3] = loaded ? staged.offsetOffsets[c] : -1;
            out[at + 4] = loaded ? staged.nullOffsets[c] : -1;
            out[at + 5] = loaded ? 1 : 0;
        }
        return out;
    }

    /** LazyBlock.getLoadedBlock of channel c of the current page, staged: the addresses of [values, offsets, nulls], 0 = absent. */
    long[] loadBlock(int channel)
    {
        Block loaded = current.getBlock(channel).getLoadedBlock();
        return staging.stageBlock(channel, loaded);
    }

    void close()
    {
        try {
            source.close();
        }
        catch (java.io.IOException e) {
            throw new java.io.UncheckedIOException(e);
        }
        finally {
            staging.close();
        }
    }
}
